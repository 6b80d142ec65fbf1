import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r_tucker_amd as rt
lib = rt._lib.load()
def run(M, N, K, ak, bk):
    rng = np.random.default_rng(1)
    A = rng.standard_normal((M, K)).astype(np.float32); Bm = rng.standard_normal((N, K)).astype(np.float32)
    dA = torch.from_numpy(A if ak else np.ascontiguousarray(A.T)).cuda()
    dB = torch.from_numpy(Bm if bk else np.ascontiguousarray(Bm.T)).cuda()
    lda, ldb = (K if ak else M), (K if bk else N)
    ba = torch.tensor([np.abs(A).max()], device="cuda"); bb = torch.tensor([np.abs(Bm).max()], device="cuda")
    C = torch.zeros(M, N, device="cuda")
    rc = lib.rtk_gemm_sf16_splitk(dA.data_ptr(), ak, lda, ba.data_ptr(), dB.data_ptr(), bk, ldb, bb.data_ptr(), C.data_ptr(), N, M, N, K, 1, None, 0, torch.cuda.current_stream().cuda_stream)
    ref = A.astype(np.float64) @ Bm.astype(np.float64).T
    err = np.abs(C.cpu().numpy() - ref)
    print((M, N, K, ak, bk), "rc", rc, "max err", err.max(), "bad rows", np.where(err.max(1) > 1e-3)[0][:10], "bad cols", np.where(err.max(0) > 1e-3)[0][:10])
    # which k contribute wrongly: compare with partial sums
    if err.max() > 1e-3:
        for kk in (16, 32, 64, 96):
            refk = A[:, :kk].astype(np.float64) @ Bm[:, :kk].astype(np.float64).T
            print("   prefix", kk, np.abs(C.cpu().numpy() - refk).max())
for cfg in [(64, 33, 100, 0, 1), (64, 33, 128, 0, 1), (64, 32, 100, 0, 1), (64, 33, 100, 1, 1), (64, 33, 100, 0, 0), (128, 128, 96, 0, 1), (64, 33, 96, 0, 1)]:
    run(*cfg)
