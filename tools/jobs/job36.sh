#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j36; mkdir -p $O
BEER_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-exact > $O/c2_g2.json 2>$O/err1.log; tail -c 400 $O/c2_g2.json; tail -3 $O/err1.log
BEER_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config 3 --frames 2000000 --steps 2 --warmup 1 --no-cpu-baseline > $O/c3_g2.json 2>$O/err2.log; tail -c 300 $O/c3_g2.json; tail -3 $O/err2.log
BEER_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-exact > $O/c2_tr.json 2>$O/err3.log; tail -c 300 $O/c2_tr.json; tail -3 $O/err3.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
