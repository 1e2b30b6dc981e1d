"""Shapes of the BASELINE.json configurations and the gfx950 peaks the rooflines are priced on."""

K, D = 256, 40                       # config 2: components, feature dimension
Q = D * D + D + 2                    # statistics per frame of a full-covariance Gaussian
# MI355X_MICROARCH.md: dense MFMA peaks (f32 operands; bf16 operands / f32 accumulate), the fp32
# vector peak (= the f32 MFMA peak), HBM3E
PEAK_TFLOPS = {'f32': 157.3, 'f64': 78.6, 'bf16': 2500.}
PEAK_HBM_GBS = 8000.
# config 3 / 4 / 5: phone loop (recipes/aud/conf/hmm.yml topology)
N_PHONES, N_COMP = 40, 16
TOPO = [(0, 1, 1.), (1, 1, .75), (1, 2, .25), (2, 2, .75), (2, 3, .25), (3, 3, .75), (3, 4, .25)]
LATENT = 64                          # config 4: dimension of the VAE's latent variable
