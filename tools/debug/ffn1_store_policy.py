import sys, os, statistics, torch
sys.path.insert(0, os.getcwd())
from musediffusion_amd import _lib
_lib.use_debug_library()
L = _lib.lib()
dev, bf = "cuda", torch.bfloat16
for M in (16384, 32768):
    N, K = 2048, 512
    X = (torch.randn(M, K, device=dev)).to(bf); W = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
    b = torch.zeros(N, device=dev); outs = [torch.empty(M, N, device=dev, dtype=bf) for _ in range(4)]
    def run(i): _lib.check(L.mh_gemm_bias_act_ex(X.data_ptr(), M, 1, W.data_ptr(), N, 1, b.data_ptr(), None, 0, 0, outs[i % 4].data_ptr(), M, 1, 0, M, N, K, 2, 1, _lib.current_stream()))
    res = {0: [], 2: []}
    for rnd in range(7):
        for ps in (0, 2):
            L.mh_gemm_set_plain_stores(ps)
            run(0); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20): run(i)
            e1.record(); torch.cuda.synchronize()
            res[ps].append(e0.elapsed_time(e1) / 20 * 1e3)
    print("FFN1 + GELU M=%d: streaming (nt) stores %.1f us, ordinary stores %.1f us" % (M, statistics.median(res[0]), statistics.median(res[2])), flush=True)
