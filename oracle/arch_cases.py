"""Non-default NeRF / Embedding configurations of the golden file tests/golden/reference_golden_arch.npz (test infrastructure;
shared by oracle/make_golden_arch.py, which mints the vectors from the real reference, and the tests that read them)."""
# tag -> (make_arch kwargs, weight seed)
ARCHS = {
    "d4w96": (dict(D=4, W=96, N_freq_xyz=6, N_freq_dir=2, skips=(2,), logscale=True), 41),
    "d3w80lin": (dict(D=3, W=80, N_freq_xyz=5, N_freq_dir=3, skips=(1, 2), logscale=False), 43),
    "d5w64": (dict(D=5, W=64, N_freq_xyz=10, N_freq_dir=4, skips=(1, 3), logscale=True), 47),     # the default embeddings
}
N_PTS, N_RAYS, S_C, N_I = 96, 24, 16, 8
