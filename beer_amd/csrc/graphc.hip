// Host-side graph compiler (no device code): removes the non-emitting states
// of an HMM topology and produces the CSR images the inference kernels read.
//
// Reference restated: Graph.compile / find_next_pdf_ids / find_previous_pdf_ids
// (beer/graph.py:156-240) and create_graph_from_seq
// (beer/cli/subcommands/hmm/mkaligraph.py:18-39).  The reference is pure Python
// whose `arcs()` scans the whole arc set for every state (graph.py:82-101), i.e.
// O(states x arcs) per graph and one interpreter round trip per arc; here a
// graph is compiled in O(states + arcs) and a whole corpus of alignment graphs
// is built from phone strings in ONE call, laid out as ONE blob that goes to
// the GPU in one copy (beer_graphset_image).
//
// Arithmetic follows the reference: path weights are products of doubles
// (Python floats), the probability tables are float32 (torch.zeros default),
// normalised in float32, and the log is taken in float32.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <utility>
#include <vector>

#include "beer_hip.h"

struct CGraph {
    int32_t S = 0;
    std::vector<float> init, fin;        // probabilities
    std::vector<int32_t> pdf;
    std::vector<int32_t> asrc, adst;     // arcs sorted by (src, dst)
    std::vector<float> aprob;
};

struct beer_graphset {
    std::vector<CGraph> graphs;
};

namespace {

// Graphs of a corpus are independent: ranges of them go to a few host threads (a corpus of
// 33 k utterances is 0.2 s of one core; the host of an MI355X has 128).
template <typename F>
void parallel_ranges(int64_t n, F&& body) {
    int nt = (int)std::thread::hardware_concurrency();
    if (const char* env = std::getenv("BEER_GRAPHC_THREADS")) nt = std::atoi(env);
    nt = nt > 16 ? 16 : (nt < 1 ? 1 : nt);
    if ((int64_t)nt > n / 128) nt = (int)(n / 128);
    if (nt <= 1) { body((int64_t)0, n); return; }
    std::vector<std::thread> pool;
    const int64_t per = (n + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        const int64_t lo = t * per, hi = lo + per < n ? lo + per : n;
        if (lo >= hi) break;
        pool.emplace_back([&body, lo, hi] { body(lo, hi); });
    }
    for (auto& th : pool) th.join();
}


struct Topology {
    int32_t n = 0;
    const int32_t* pdf = nullptr;
    std::vector<int32_t> src, dst;
    std::vector<double> w;
    int32_t start = 0, end = 0;
};

struct Walker {
    // adjacency in arc-insertion order
    std::vector<int32_t> out_ptr, out_arc, in_ptr, in_arc;
    std::vector<int32_t> stamp;
    int32_t tick = 0;
    std::vector<std::pair<int32_t, double>> stack;

    void build(const Topology& g) {
        const size_t m = g.src.size();
        out_ptr.assign(g.n + 1, 0);
        in_ptr.assign(g.n + 1, 0);
        for (size_t a = 0; a < m; ++a) {
            ++out_ptr[g.src[a] + 1];
            ++in_ptr[g.dst[a] + 1];
        }
        for (int i = 0; i < g.n; ++i) {
            out_ptr[i + 1] += out_ptr[i];
            in_ptr[i + 1] += in_ptr[i];
        }
        out_arc.resize(m);
        in_arc.resize(m);
        std::vector<int32_t> po(out_ptr.begin(), out_ptr.end() - 1), pi(in_ptr.begin(),
                                                                         in_ptr.end() - 1);
        for (size_t a = 0; a < m; ++a) {
            out_arc[po[g.src[a]]++] = (int32_t)a;
            in_arc[pi[g.dst[a]]++] = (int32_t)a;
        }
        stamp.assign(g.n, 0);
    }

    // graph.py:156-182: follow non-emitting states (each expanded once per walk)
    // until emitting ones; f(state, weight) for every emitting state reached.
    template <typename F>
    void walk(const Topology& g, int32_t from, double w0, bool incoming, F&& f) {
        ++tick;
        stack.clear();
        const auto& ptr = incoming ? in_ptr : out_ptr;
        const auto& arc = incoming ? in_arc : out_arc;
        for (int32_t k = ptr[from]; k < ptr[from + 1]; ++k) stack.emplace_back(arc[k], w0);
        stamp[from] = tick;
        while (!stack.empty()) {
            const auto [a, w] = stack.back();
            stack.pop_back();
            const int32_t nxt = incoming ? g.src[a] : g.dst[a];
            if (g.pdf[nxt] >= 0) {
                f(nxt, w * g.w[a]);
            } else if (stamp[nxt] != tick) {
                for (int32_t k = ptr[nxt]; k < ptr[nxt + 1]; ++k)
                    stack.emplace_back(arc[k], g.w[a] * w);
                stamp[nxt] = tick;
            }
        }
    }
};

float fsum(const std::vector<float>& v) {
    double s = 0.0;
    for (float x : v) s += (double)x;
    return (float)s;
}

// graph.py:185-240
int compile_topology(const Topology& g, Walker& wk, CGraph& out) {
    std::vector<int32_t> index(g.n, -1);
    out.pdf.clear();
    for (int i = 0; i < g.n; ++i)
        if (g.pdf[i] >= 0) {
            index[i] = (int32_t)out.pdf.size();
            out.pdf.push_back(g.pdf[i]);
        }
    const int32_t S = (int32_t)out.pdf.size();
    out.S = S;
    out.init.assign(S, 0.f);
    out.fin.assign(S, 0.f);
    wk.build(g);
    wk.walk(g, g.start, 1.0, false, [&](int32_t s, double w) { out.init[index[s]] += (float)w; });
    const float si = fsum(out.init);
    for (float& x : out.init) x /= si;
    wk.walk(g, g.end, 1.0, true, [&](int32_t s, double w) { out.fin[index[s]] += (float)w; });
    const float sf = fsum(out.fin);
    for (float& x : out.fin) x /= sf;

    // transition entries, gathered per source row
    std::vector<std::vector<std::pair<int32_t, float>>> rows(S);
    auto add = [&](int32_t r, int32_t c, double w) {
        for (auto& e : rows[r])
            if (e.first == c) {
                e.second += (float)w;
                return;
            }
        rows[r].emplace_back(c, (float)w);
    };
    const size_t m = g.src.size();
    for (size_t a = 0; a < m; ++a) {
        if (g.pdf[g.src[a]] < 0) continue;                    // handled by init
        const int32_t r = index[g.src[a]];
        if (g.pdf[g.dst[a]] < 0)
            wk.walk(g, g.dst[a], g.w[a], false,
                    [&](int32_t s, double w) { add(r, index[s], w); });
        else
            add(r, index[g.dst[a]], g.w[a]);
    }
    out.asrc.clear();
    out.adst.clear();
    out.aprob.clear();
    for (int32_t r = 0; r < S; ++r) {
        auto& row = rows[r];
        std::sort(row.begin(), row.end());
        // renormalise without changing the self-loop probability (graph.py:230-237)
        float diag = 0.f;
        double tot = 0.0;
        for (auto& e : row) {
            if (e.first == r) diag = e.second;
            tot += (double)e.second;
        }
        const float off = (float)tot - diag;
        if (diag > 0.f && off > 0.f) {
            const float q = off / (1.f - diag);
            for (auto& e : row)
                if (e.first != r) e.second /= q;
        }
        for (auto& e : row) {
            if (!(e.second > 0.f)) continue;                 // log(0) = -inf: no arc
            out.asrc.push_back(r);
            out.adst.push_back(e.first);
            out.aprob.push_back(e.second);
        }
    }
    return BEER_OK;
}

size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

struct Layout {
    size_t init, fin, in_ptr, in_src, in_dst, in_w, in_seg, in_row_seg;
    size_t out_ptr, out_dst, out_src, out_w, out_seg, out_row_seg;
    size_t hub_id, hub_w, hub_ptr, lowdeg, end;
    int32_t n_in_seg, n_out_seg;
    bool lowdeg_ok;
};

int32_t count_segs(const std::vector<int32_t>& ptr) {
    int32_t n = 0;
    for (size_t r = 0; r + 1 < ptr.size(); ++r) n += (ptr[r + 1] - ptr[r] + BEER_SEG - 1) / BEER_SEG;
    return n;
}

void csr_ptr(const std::vector<int32_t>& key, int32_t S, std::vector<int32_t>& ptr) {
    ptr.assign(S + 1, 0);
    for (int32_t k : key) ++ptr[k + 1];
    for (int32_t i = 0; i < S; ++i) ptr[i + 1] += ptr[i];
}

}  // namespace

extern "C" {

int beer_graph_compile(int32_t n_states, const int32_t* pdf_ids, int64_t n_arcs,
                       const int32_t* arc_src, const int32_t* arc_dst, const double* arc_w,
                       int32_t start_state, int32_t end_state, beer_graphset** out) {
    if (!out || n_states < 2 || !pdf_ids || n_arcs < 0 || (n_arcs && (!arc_src || !arc_dst || !arc_w)))
        return BEER_EINVAL;
    if (start_state < 0 || start_state >= n_states || end_state < 0 || end_state >= n_states)
        return BEER_EINVAL;
    for (int64_t a = 0; a < n_arcs; ++a)
        if (arc_src[a] < 0 || arc_src[a] >= n_states || arc_dst[a] < 0 || arc_dst[a] >= n_states)
            return BEER_EINVAL;
    Topology g;
    g.n = n_states;
    g.pdf = pdf_ids;
    g.src.assign(arc_src, arc_src + n_arcs);
    g.dst.assign(arc_dst, arc_dst + n_arcs);
    g.w.assign(arc_w, arc_w + n_arcs);
    g.start = start_state;
    g.end = end_state;
    auto* set = new (std::nothrow) beer_graphset;
    if (!set) return BEER_EINVAL;
    set->graphs.resize(1);
    Walker wk;
    const int rc = compile_topology(g, wk, set->graphs[0]);
    if (rc != BEER_OK) {
        delete set;
        return rc;
    }
    *out = set;
    return BEER_OK;
}

int beer_aligraphs_compile(int32_t n_units, const int32_t* unit_state_off,
                           const int32_t* unit_pdf_ids, const int32_t* unit_start,
                           const int32_t* unit_end, const int32_t* unit_arc_off,
                           const int32_t* unit_arc_src, const int32_t* unit_arc_dst,
                           const double* unit_arc_w, int64_t n_utts, const int64_t* seq_off,
                           const int32_t* seq_units, beer_graphset** out) {
    if (!out || n_units < 1 || !unit_state_off || !unit_pdf_ids || !unit_start || !unit_end ||
        !unit_arc_off || n_utts < 0 || !seq_off)
        return BEER_EINVAL;
    for (int64_t i = 0; i < seq_off[n_utts]; ++i)
        if (seq_units[i] < 0 || seq_units[i] >= n_units) return BEER_EINVAL;
    auto* set = new (std::nothrow) beer_graphset;
    if (!set) return BEER_EINVAL;
    set->graphs.resize(n_utts);
    for (int64_t u = 0; u < n_utts; ++u)
        if (seq_off[u + 1] - seq_off[u] < 1) {
            delete set;
            return BEER_EINVAL;
        }
    int failed = BEER_OK;
    parallel_ranges(n_utts, [&](int64_t u_lo, int64_t u_hi) {
    Walker wk;
    Topology g;
    std::vector<int32_t> pdf;
    std::vector<double> total;
    for (int64_t u = u_lo; u < u_hi; ++u) {
        const int64_t n = seq_off[u + 1] - seq_off[u];
        const int32_t* seq = seq_units + seq_off[u];
        // states in the order the reference's OrderedDict ends up with: start,
        // end, then the states of every unit copy (mkaligraph.py:18-39)
        pdf.assign(2, -1);
        g.src.clear();
        g.dst.clear();
        g.w.clear();
        int32_t last_end = 0;                              // the start state
        for (int64_t i = 0; i < n; ++i) {
            const int32_t p = seq[i];
            const int32_t s0 = unit_state_off[p], ns = unit_state_off[p + 1] - s0;
            const int32_t base = (int32_t)pdf.size();
            for (int32_t k = 0; k < ns; ++k) pdf.push_back(unit_pdf_ids[s0 + k]);
            for (int32_t a = unit_arc_off[p]; a < unit_arc_off[p + 1]; ++a) {
                g.src.push_back(base + unit_arc_src[a]);
                g.dst.push_back(base + unit_arc_dst[a]);
                g.w.push_back(unit_arc_w[a]);
            }
            g.src.push_back(last_end);
            g.dst.push_back(base + unit_start[p]);
            g.w.push_back(1.0);
            last_end = base + unit_end[p];
        }
        g.src.push_back(last_end);
        g.dst.push_back(1);                                // the end state
        g.w.push_back(1.0);
        g.n = (int32_t)pdf.size();
        g.pdf = pdf.data();
        g.start = 0;
        g.end = 1;
        // Graph.normalize (graph.py:126-135)
        total.assign(g.n, 0.0);
        for (size_t a = 0; a < g.src.size(); ++a) total[g.src[a]] += g.w[a];
        for (size_t a = 0; a < g.src.size(); ++a) g.w[a] /= total[g.src[a]];
        const int rc = compile_topology(g, wk, set->graphs[u]);
        if (rc != BEER_OK) {
            __atomic_store_n(&failed, rc, __ATOMIC_RELAXED);
            return;
        }
    }
    });
    if (failed != BEER_OK) {
        delete set;
        return failed;
    }
    *out = set;
    return BEER_OK;
}

int beer_graphset_free(beer_graphset* set) {
    delete set;
    return BEER_OK;
}

int beer_graphset_sizes(const beer_graphset* set, int64_t* n_graphs, int64_t* state_off,
                        int64_t* arc_off) {
    if (!set || !n_graphs) return BEER_EINVAL;
    *n_graphs = (int64_t)set->graphs.size();
    if (state_off && arc_off) {
        state_off[0] = arc_off[0] = 0;
        for (size_t i = 0; i < set->graphs.size(); ++i) {
            state_off[i + 1] = state_off[i] + set->graphs[i].S;
            arc_off[i + 1] = arc_off[i] + (int64_t)set->graphs[i].asrc.size();
        }
    }
    return BEER_OK;
}

int beer_graphset_export(const beer_graphset* set, float* init, float* fin, int32_t* pdf_ids,
                         int32_t* arc_src, int32_t* arc_dst, float* arc_prob) {
    if (!set) return BEER_EINVAL;
    size_t so = 0, ao = 0;
    for (const CGraph& g : set->graphs) {
        if (init) std::copy(g.init.begin(), g.init.end(), init + so);
        if (fin) std::copy(g.fin.begin(), g.fin.end(), fin + so);
        if (pdf_ids) std::copy(g.pdf.begin(), g.pdf.end(), pdf_ids + so);
        if (arc_src) std::copy(g.asrc.begin(), g.asrc.end(), arc_src + ao);
        if (arc_dst) std::copy(g.adst.begin(), g.adst.end(), arc_dst + ao);
        if (arc_prob) std::copy(g.aprob.begin(), g.aprob.end(), arc_prob + ao);
        so += g.S;
        ao += g.asrc.size();
    }
    return BEER_OK;
}

}  // extern "C"

namespace {

template <typename T>
Layout layout_of(const CGraph& g, size_t base) {
    Layout L;
    const size_t S = g.S, A = g.asrc.size();
    std::vector<int32_t> ip, op;
    csr_ptr(g.adst, g.S, ip);
    csr_ptr(g.asrc, g.S, op);
    L.n_in_seg = count_segs(ip);
    L.n_out_seg = count_segs(op);
    L.lowdeg_ok = true;
    for (size_t r = 0; r < S; ++r)
        if (ip[r + 1] - ip[r] > BEER_SEG || op[r + 1] - op[r] > BEER_SEG) L.lowdeg_ok = false;
    size_t o = align16(base);
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o = align16(o + bytes);
        return at;
    };
    L.init = take(S * sizeof(T));
    L.fin = take(S * sizeof(T));
    L.in_ptr = take((S + 1) * 4);
    L.in_src = take(A * 4);
    L.in_dst = take(A * 4);
    L.in_w = take(A * sizeof(T));
    L.in_seg = take((L.n_in_seg + 1) * 4);
    L.in_row_seg = take((S + 1) * 4);
    L.out_ptr = take((S + 1) * 4);
    L.out_dst = take(A * 4);
    L.out_src = take(A * 4);
    L.out_w = take(A * sizeof(T));
    L.out_seg = take((L.n_out_seg + 1) * 4);
    L.out_row_seg = take((S + 1) * 4);
    L.hub_id = take(S * 4);                    // all -1 (no hub)
    L.hub_w = take(S * sizeof(T));             // zeros
    L.hub_ptr = take(8);                       // [0] and a dummy list entry
    L.lowdeg = take(sizeof(beer_graph_lowdeg));
    L.end = o;
    return L;
}

void segments(const std::vector<int32_t>& ptr, int32_t* seg, int32_t* row_seg) {
    int32_t n = 0;
    row_seg[0] = 0;
    for (size_t r = 0; r + 1 < ptr.size(); ++r) {
        for (int32_t b = ptr[r]; b < ptr[r + 1]; b += BEER_SEG) seg[n++] = b;
        row_seg[r + 1] = n;
    }
    seg[n] = ptr.back();
}

template <typename T>
int image(const beer_graphset* set, char* blob, uint64_t dev, beer_graph* structs) {
    // where every graph's image starts (a prefix sum), then the images themselves in parallel
    std::vector<size_t> bases(set->graphs.size() + 1, 0);
    for (size_t gi = 0; gi < set->graphs.size(); ++gi)
        bases[gi + 1] = layout_of<T>(set->graphs[gi], bases[gi]).end;
    parallel_ranges((int64_t)set->graphs.size(), [&](int64_t g_lo, int64_t g_hi) {
    for (size_t gi = (size_t)g_lo; gi < (size_t)g_hi; ++gi) {
        const CGraph& g = set->graphs[gi];
        const Layout L = layout_of<T>(g, bases[gi]);
        const int32_t S = g.S;
        const size_t A = g.asrc.size();
        T* init = (T*)(blob + L.init);
        T* fin = (T*)(blob + L.fin);
        for (int32_t s = 0; s < S; ++s) {
            init[s] = (T)logf(g.init[s]);
            fin[s] = (T)logf(g.fin[s]);
        }
        // by source (arcs are stored sorted by (src, dst))
        std::vector<int32_t> op, ip;
        csr_ptr(g.asrc, S, op);
        csr_ptr(g.adst, S, ip);
        std::memcpy(blob + L.out_ptr, op.data(), (S + 1) * 4);
        std::memcpy(blob + L.out_dst, g.adst.data(), A * 4);
        std::memcpy(blob + L.out_src, g.asrc.data(), A * 4);
        T* ow = (T*)(blob + L.out_w);
        for (size_t a = 0; a < A; ++a) ow[a] = (T)logf(g.aprob[a]);
        segments(op, (int32_t*)(blob + L.out_seg), (int32_t*)(blob + L.out_row_seg));
        // by destination, sources ascending (stable counting sort of the above)
        std::memcpy(blob + L.in_ptr, ip.data(), (S + 1) * 4);
        std::vector<int32_t> cur(ip.begin(), ip.end() - 1);
        int32_t* isrc = (int32_t*)(blob + L.in_src);
        int32_t* idst = (int32_t*)(blob + L.in_dst);
        T* iw = (T*)(blob + L.in_w);
        for (size_t a = 0; a < A; ++a) {
            const int32_t k = cur[g.adst[a]]++;
            isrc[k] = g.asrc[a];
            idst[k] = g.adst[a];
            iw[k] = ow[a];
        }
        segments(ip, (int32_t*)(blob + L.in_seg), (int32_t*)(blob + L.in_row_seg));
        int32_t* hub_id = (int32_t*)(blob + L.hub_id);
        for (int32_t s = 0; s < S; ++s) hub_id[s] = -1;
        std::memset(blob + L.hub_w, 0, S * sizeof(T));
        std::memset(blob + L.hub_ptr, 0, 8);
        auto d = [&](size_t off) { return (const void*)(uintptr_t)(dev + off); };
        beer_graph_lowdeg ld;
        ld.n_arcs = (int32_t)A;
        ld.n_hubs = 0;
        ld.in_ptr = (const int32_t*)d(L.in_ptr);
        ld.in_src = (const int32_t*)d(L.in_src);
        ld.in_w = d(L.in_w);
        ld.out_ptr = (const int32_t*)d(L.out_ptr);
        ld.out_dst = (const int32_t*)d(L.out_dst);
        ld.out_w = d(L.out_w);
        ld.hub_src_id = ld.hub_dst_id = (const int32_t*)d(L.hub_id);
        ld.hub_src_w = ld.hub_dst_w = d(L.hub_w);
        ld.src_ptr = ld.dst_ptr = (const int32_t*)d(L.hub_ptr);
        ld.src_list = ld.dst_list = (const int32_t*)d(L.hub_ptr + 4);
        std::memcpy(blob + L.lowdeg, &ld, sizeof(ld));
        beer_graph& st = structs[gi];
        st.n_states = S;
        st.n_arcs = (int32_t)A;
        st.n_in_seg = L.n_in_seg;
        st.n_out_seg = L.n_out_seg;
        st.init = d(L.init);
        st.final = d(L.fin);
        st.in_ptr = (const int32_t*)d(L.in_ptr);
        st.in_src = (const int32_t*)d(L.in_src);
        st.in_dst = (const int32_t*)d(L.in_dst);
        st.in_w = d(L.in_w);
        st.in_seg = (const int32_t*)d(L.in_seg);
        st.in_row_seg = (const int32_t*)d(L.in_row_seg);
        st.out_ptr = (const int32_t*)d(L.out_ptr);
        st.out_dst = (const int32_t*)d(L.out_dst);
        st.out_src = (const int32_t*)d(L.out_src);
        st.out_w = d(L.out_w);
        st.out_seg = (const int32_t*)d(L.out_seg);
        st.out_row_seg = (const int32_t*)d(L.out_row_seg);
        st.lowdeg = L.lowdeg_ok ? (const beer_graph_lowdeg*)d(L.lowdeg) : nullptr;
    }
    });
    return BEER_OK;
}

}  // namespace

extern "C" {

int beer_graphset_image_bytes(const beer_graphset* set, int dtype, size_t* bytes) {
    if (!set || !bytes || (dtype != BEER_F32 && dtype != BEER_F64)) return BEER_EINVAL;
    size_t base = 0;
    for (const CGraph& g : set->graphs)
        base = dtype == BEER_F32 ? layout_of<float>(g, base).end : layout_of<double>(g, base).end;
    *bytes = base;
    return BEER_OK;
}

int beer_graphset_image(const beer_graphset* set, int dtype, void* host_blob,
                        uint64_t device_base, beer_graph* graphs) {
    if (!set || !host_blob || !graphs) return BEER_EINVAL;
    if (dtype == BEER_F32) return image<float>(set, (char*)host_blob, device_base, graphs);
    if (dtype == BEER_F64) return image<double>(set, (char*)host_blob, device_base, graphs);
    return BEER_EINVAL;
}

}  // extern "C"
