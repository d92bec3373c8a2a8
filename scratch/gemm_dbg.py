import os, sys, torch
sys.path.insert(0, os.getcwd())
from leclip_amd.hip import ops
torch.manual_seed(0)
dev = "cuda"
for dt in (torch.bfloat16, torch.float16):
    for (M, N, K) in ((6160, 2048, 512), (6144, 2048, 512), (50432, 768, 768)):
        a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
        bias = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev).to(dt)
        stats = torch.stack([torch.randn(M, device=dev) * 0.1, torch.rand(M, device=dev) + 0.5], dim=1).contiguous()
        colsum = torch.randn(N, device=dev)
        ref0 = a.float() @ w.float().t()
        def chk(name, y, ref):
            err = (y.float() - ref).abs().max().item(); print(dt, (M, N, K), name, "max err", err, "ref max", ref.abs().max().item(), flush=True)
        chk("bias", ops.gemm(a, w, bias), ref0 + bias)
        chk("bias+gelu", ops.gemm(a, w, bias, act=ops.ACT_QUICKGELU), (lambda x: x * torch.sigmoid(1.702 * x))(ref0 + bias))
        chk("res", ops.gemm(a, w, bias, residual=res), ref0 + bias + res.float())
        so = torch.zeros(M, N // 64, 2, device=dev)
        y = ops.gemm_ln(a, w, bias, residual=res, stats_out=so)
        chk("res+stats", y, ref0 + bias + res.float())
        yy = y.float().view(M, N // 64, 64)
        print("   stats err", (so[..., 0] - yy.sum(-1)).abs().max().item(), (so[..., 1] - (yy * yy).sum(-1)).abs().max().item())
        lnref = stats[:, 1:2] * (ref0 - stats[:, 0:1] * colsum[None, :]) + bias
        chk("ln", ops.gemm_ln(a, w, bias, ln_stats=stats, ln_colsum=colsum), lnref)
        chk("ln+gelu", ops.gemm_ln(a, w, bias, ln_stats=stats, ln_colsum=colsum, act=ops.ACT_QUICKGELU), (lambda x: x * torch.sigmoid(1.702 * x))(lnref))
