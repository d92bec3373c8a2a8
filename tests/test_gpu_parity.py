"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the committed golden vectors.

Tolerances (north-star: "within 1e-3 fp32, label-index outputs bit-exact"):
* fp32 mode  - logits <= 1e-3 abs against the golden vectors of the reference forward, top-5 indices exact;
* fp16/bf16  - logits within the measured low-precision band (fp16 3e-3, bf16 3e-2 at scale 4.0), argmax exact
  on samples whose oracle top-1 margin exceeds twice that band.
"""
import os

import numpy as np
import pytest
import torch

from leclip_amd import synth

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
DTYPES = [torch.float32, torch.float16, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from leclip_amd.hip import ops as _ops, _capi
    _capi.load()  # fails loudly if the library is missing
    return _ops


def _rand(shape, seed, std=1.0):
    return torch.from_numpy(synth.normal(seed, "t", shape, std=std))


def _tol(dt, f32=2e-5, f16=4e-3, bf16=3e-2):
    return {torch.float32: f32, torch.float16: f16, torch.bfloat16: bf16}[dt]


# ----------------------------------------------------------------------------------------------- per-op parity
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("dim", [64, 128, 512, 768, 1024])
def test_layernorm(ops, dt, dim):
    from oracle import clip_oracle as co
    x = _rand((37, dim), 1, 3.0) + 0.5
    g, b = _rand((dim,), 2) * 0.1 + 1, _rand((dim,), 3) * 0.1
    xq = x.to(dt)
    ref = co.layer_norm(xq.float(), g, b)
    for odt in (dt, torch.float32):
        y = ops.layernorm(xq.to(DEV), g.to(DEV), b.to(DEV), out_dtype=odt)
        assert y.dtype == odt
        np.testing.assert_allclose(y.float().cpu().numpy(), ref.numpy(), atol=_tol(odt, 2e-5, 4e-3, 3e-2), rtol=0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(1, 128, 64), (197, 768, 768), (1000, 256, 3072), (130, 384, 192), (49300, 256, 128)])
@pytest.mark.parametrize("epi", ["plain", "bias_gelu", "bias_res"])
def test_gemm(ops, dt, shape, epi):
    from leclip_amd.hip import ops as o
    m, n, k = shape
    a, w = _rand((m, k), 4).to(dt), _rand((n, k), 5, k ** -0.5).to(dt)
    bias, res = _rand((n,), 6), _rand((m, n), 7).to(dt)
    ref = a.double() @ w.double().t()
    kw = {}
    if epi != "plain":
        ref = ref + bias.double()
        kw["bias"] = bias.to(DEV)
    if epi == "bias_gelu":
        ref = ref * torch.sigmoid(1.702 * ref)
        kw["act"] = o.ACT_QUICKGELU
    if epi == "bias_res":
        ref = ref + res.double()
        kw["residual"] = res.to(DEV)
    y = ops.gemm(a.to(DEV), w.to(DEV), **kw)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(y.double().cpu().numpy(), ref.numpy(), atol=_tol(dt, 2e-6, 2e-3, 1.2e-2) * scale, rtol=0)
    y32 = ops.gemm(a.to(DEV), w.to(DEV), out_dtype=torch.float32, **kw)
    np.testing.assert_allclose(y32.double().cpu().numpy(), ref.numpy(), atol=_tol(dt, 2e-6, 1e-3, 6e-3) * scale, rtol=0)


def test_gemm_kernel_families(ops):
    """The dispatcher: the 384x256 ping-pong kernel (round 5) from 160 of its tiles up, the 256x256 one from 96 of its tiles, 128x128 otherwise;
    against fp64."""
    from leclip_amd.hip import _capi
    if os.environ.get("LECLIP_TEST_GEMM_FAMILY"):
        pytest.skip("the default dispatch is not in effect under LECLIP_TEST_GEMM_FAMILY")
    lib = _capi.load()
    assert lib.leclip_gemm_kernel_name(50432, 768, 768, _capi.BF16) == b"gemm_tn_384x256x32_pp"
    assert lib.leclip_gemm_kernel_name(12608, 768, 768, _capi.BF16) == b"gemm_tn_256x256x64_pp"      # 99 tiles of 384x256, 150 of 256x256
    assert lib.leclip_gemm_kernel_name(1576, 768, 768, _capi.BF16) == b"gemm_tn_128x128x64"
    assert lib.leclip_gemm_kernel_name(50432, 768, 768, _capi.F32) == b"gemm_f32_64x64x32"
    assert lib.leclip_gemm_kernel_name(50432, 100, 768, _capi.BF16) == b"unsupported"
    # ragged M through the persistent kernel (multi-tile loop per workgroup: 197*3 tiles on 256 CUs)
    m, n, k = 50432 - 37, 768, 192
    a, w = _rand((m, k), 31).bfloat16(), _rand((n, k), 32, k ** -0.5).bfloat16()
    y = ops.gemm(a.to(DEV), w.to(DEV), out_dtype=torch.float32)
    ref = a.double() @ w.double().t()
    np.testing.assert_allclose(y.double().cpu().numpy(), ref.numpy(), atol=2e-3 * float(ref.abs().max()), rtol=0)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(300, 256, 128), (1576, 2304, 768), (49300, 256, 128)])
def test_gemm_with_fused_layernorm(ops, dt, shape):
    """leclip_gemm_ln_fused_fwd: LayerNorm folded around the GEMM (gamma into W, mean/rstd in the epilogue) against
    LN -> linear in fp64, and the output-row statistics it emits for the next LayerNorm (128 and 256 kernel families)."""
    from leclip_amd.hip import engine, ops as o
    m, n, k = shape
    x = (_rand((m, k), 41, 2.0) + 0.7).to(dt)
    gamma, beta = _rand((k,), 42) * 0.1 + 1, _rand((k,), 43) * 0.1
    w, b = _rand((n, k), 44, k ** -0.5), _rand((n,), 45) * 0.1
    res = _rand((m, n), 46).to(dt)
    xd = x.double()
    mu, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    ref = ((xd - mu) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()) @ w.double().t() + b.double()
    ref_act = ref * torch.sigmoid(1.702 * ref) + res.double()
    wf, cs, cb = engine._fold_ln(w.to(DEV), b.to(DEV), gamma.to(DEV), beta.to(DEV), dt)
    stats = ops.row_stats(x.to(DEV))
    np.testing.assert_allclose(stats[:, 0].cpu().numpy(), mu[:, 0].numpy(), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(stats[:, 1].cpu().numpy(), (1 / torch.sqrt(var + 1e-5))[:, 0].numpy(), rtol=1e-4)
    y = ops.gemm_ln(x.to(DEV), wf, cb, ln_stats=stats, ln_colsum=cs, out_dtype=torch.float32)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(y.double().cpu().numpy(), ref.numpy(), atol=_tol(dt, 0, 2.5e-3, 1.5e-2) * scale, rtol=0)
    part = torch.empty((n // 64, m, 2), dtype=torch.float32, device=DEV)      # slot-major [N/64][M][2]
    y2 = ops.gemm_ln(x.to(DEV), wf, cb, ln_stats=stats, ln_colsum=cs, residual=res.to(DEV), act=o.ACT_QUICKGELU, stats_out=part)
    np.testing.assert_allclose(y2.double().cpu().numpy(), ref_act.numpy(), atol=_tol(dt, 0, 4e-3, 2.5e-2) * float(ref_act.abs().max()), rtol=0)
    st2 = ops.ln_stats_finalize(part, n)
    yd = y2.double().cpu()
    np.testing.assert_allclose(st2[:, 0].cpu().numpy(), yd.mean(1).numpy(), atol=2e-5 * float(yd.abs().max()), rtol=1e-4)
    np.testing.assert_allclose(st2[:, 1].cpu().numpy(), (1 / torch.sqrt(yd.var(1, unbiased=False) + 1e-5)).numpy(), rtol=2e-4)
    again = ops.gemm_ln(x.to(DEV), wf, cb, ln_stats=stats, ln_colsum=cs, residual=res.to(DEV), act=o.ACT_QUICKGELU, stats_out=part)
    assert torch.equal(again, y2) and torch.equal(ops.ln_stats_finalize(part, n), st2)      # deterministic, no atomics


def test_gemm_inplace_residual_and_errors(ops):
    from leclip_amd.hip._capi import HipKernelError
    a, w = _rand((300, 128), 8).bfloat16().to(DEV), _rand((128, 128), 9, 0.1).bfloat16().to(DEV)
    x = _rand((300, 128), 10).bfloat16().to(DEV)
    ref = ops.gemm(a, w, residual=x)
    out = ops.gemm(a, w, residual=x, out=x)
    assert out.data_ptr() == x.data_ptr() and torch.equal(out, ref)
    with pytest.raises(HipKernelError):     # N not a multiple of 128 for 16-bit operands
        ops.gemm(a, _rand((100, 128), 1).bfloat16().to(DEV))
    with pytest.raises(RuntimeError):       # CPU tensor: no fallback
        ops.gemm(a.cpu(), w.cpu())


@pytest.fixture
def family(ops, request):
    """Runs a test under a GEMM kernel-family override (leclip_set_gemm_family; thread-local, restored afterwards)."""
    prev = ops.set_gemm_family(request.param)
    yield request.param
    ops.set_gemm_family(prev)


@pytest.mark.parametrize("family", [256, 384], indirect=True)
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(6160, 2048, 512), (50332, 768, 768), (50000, 512, 320), (49000, 512, 128)])
def test_gemm256_specialised_epilogues(ops, dt, shape, family):
    """Both persistent kernel families (256x256x64 and, round 5, 384x256x32: ring of three 32-deep steps - K = 320 / 512 / 768 / 128 are 10 / 16 / 24 /
    4 steps, every remainder mod 3; fp32-output calls fall through to the 256x256 kernel there).
    Shapes with >= 192 tiles of 256x256 run the persistent kernel; each of its compile-time epilogues (bias, +QuickGELU,
    +residual, +residual+LayerNorm partial sums, fused LayerNorm, fused LayerNorm+QuickGELU) against fp32 torch on the
    device, with a ragged last tile row (M % 256 != 0) and, for the second shape, a third round that fills a third of the CUs;
    K = 320 has an odd number of K-tiles (no cross-tile pipelining: prologue between tiles), K = 128 the minimum of two."""
    M, N, K = shape
    g = torch.Generator(device="cpu").manual_seed(7)
    a = torch.randn(M, K, generator=g).to(dt).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(dt).to(DEV)
    stats = torch.stack([torch.randn(M, generator=g) * 0.1, torch.rand(M, generator=g) + 0.5], dim=1).contiguous().to(DEV)
    colsum = torch.randn(N, generator=g).to(DEV)
    assert ops._capi.load().leclip_gemm_kernel_name(M, N, K, ops.dtype_code(dt)).decode().startswith("gemm_tn_%dx256" % family)
    ref0 = a.float() @ w.float().t()
    gelu = lambda x: x * torch.sigmoid(1.702 * x)
    lnref = stats[:, 1:2] * (ref0 - stats[:, 0:1] * colsum[None, :]) + bias
    tol = (2.0e-2 if dt == torch.bfloat16 else 2.5e-3)       # relative to the largest output magnitude (one 16-bit rounding)

    def check(y, ref):
        assert float((y.float() - ref).abs().max()) <= tol * max(float(ref.abs().max()), 1.0)

    check(ops.gemm(a, w), ref0)
    check(ops.gemm(a, w, bias), ref0 + bias)
    check(ops.gemm(a, w, bias, act=ops.ACT_QUICKGELU), gelu(ref0 + bias))
    check(ops.gemm(a, w, bias, residual=res), ref0 + bias + res.float())
    part = torch.zeros(N // 64, M, 2, device=DEV)
    y = ops.gemm_ln(a, w, bias, residual=res, stats_out=part)
    check(y, ref0 + bias + res.float())
    yy = y.float().view(M, N // 64, 64)
    dev2 = ((yy - yy.mean(-1, keepdim=True)) ** 2).sum(-1)     # partials: (sum, M2 about the block mean) per 64-column block, slot-major
    pr = part.permute(1, 0, 2)
    assert float((pr[..., 0] - yy.sum(-1)).abs().max()) <= 1e-3 and float((pr[..., 1] - dev2).abs().max()) <= 2e-2
    check(ops.gemm_ln(a, w, bias, ln_stats=stats, ln_colsum=colsum), lnref)
    check(ops.gemm_ln(a, w, bias, ln_stats=stats, ln_colsum=colsum, act=ops.ACT_QUICKGELU), gelu(lnref))


@pytest.mark.parametrize("family", [256, 384], indirect=True)
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gemm256_edge_tiles_write_nothing_past_m(ops, dt, family):
    """The specialised epilogues store through buffer descriptors that end at the last valid row; the hardware must drop what an edge tile
    (M % 256 != 0) holds past it - for every flavour, the residual ones included (round 4 moved them to the same stores, and their
    residual loads and LayerNorm-partial stores to descriptors of their own).  Outputs are views of larger buffers here whose tail rows hold a
    sentinel: every flavour must leave them untouched, bit for bit.  (Row offsets travel in the bounds-checked vector offset; round 3's
    library, which carried them in the scalar offset, passes this test as well.)"""
    M, N, K = 6160, 2048, 512            # 24 full tile rows + 16 valid rows in the last one; 200 tiles: one per workgroup (384x256: 16 + 16 rows, 136 tiles)
    g = torch.Generator(device="cpu").manual_seed(11)
    a = torch.randn(M, K, generator=g).to(dt).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    stats = torch.stack([torch.randn(M, generator=g) * 0.1, torch.rand(M, generator=g) + 0.5], dim=1).contiguous().to(DEV)
    colsum = torch.randn(N, generator=g).to(DEV)
    pad = 300
    sentinel = 1234.0

    def guarded():
        buf = torch.full((M + pad, N), sentinel, dtype=dt, device=DEV)
        return buf, buf[:M]

    def untouched(buf):
        return bool((buf[M:] == sentinel).all())

    buf, out = guarded()
    ops.gemm(a, w, bias, out=out)
    assert untouched(buf), "plain epilogue wrote past M"
    buf, out = guarded()
    ops.gemm(a, w, bias, act=ops.ACT_QUICKGELU, out=out)
    assert untouched(buf), "QuickGELU epilogue wrote past M"
    buf, out = guarded()
    ops.gemm_ln(a, w, bias, ln_stats=stats, ln_colsum=colsum, out=out)
    assert untouched(buf), "fused-LayerNorm epilogue wrote past M"
    # residual flavours, in place (the residual stream x is updated by out-proj / c_proj): x lives in the guarded buffer too
    buf, out = guarded()
    res0 = torch.randn(M, N, generator=g).to(dt).to(DEV)
    out.copy_(res0)
    ops.gemm(a, w, bias, residual=out, out=out)
    assert untouched(buf), "residual epilogue wrote past M"
    ref = (a.float() @ w.float().t() + bias).to(dt).float() + res0.float()
    assert float((out.float() - ref).abs().max()) <= (6e-2 if dt == torch.bfloat16 else 8e-3) * float(ref.abs().max())
    buf, out = guarded()
    out.copy_(res0)
    part_buf = torch.full((N // 64, M, 2), sentinel, dtype=torch.float32, device=DEV)
    ops.gemm_ln(a, w, bias, residual=out, stats_out=part_buf, out=out)
    assert untouched(buf), "residual + partials epilogue wrote past M"
    yy = out.float().view(M, N // 64, 64)
    assert float((part_buf.permute(1, 0, 2)[..., 0] - yy.sum(-1)).abs().max()) <= 1e-3     # every (slot, row) pair written, none from a row past M


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gemm_families_are_bit_identical(ops, dt):
    """Which GEMM kernel family a call takes depends on M (256x256 persistent kernel when the grid fills the chip, 128x128
    otherwise).  Both feed v_mfma_f32_16x16x32 the same ascending K sequence and share the epilogue arithmetic, so a
    row's result is the same bits in a large batch and in a small one - the property behind shard / batch invariance of
    the logits.  Every epilogue flavour: plain, +bias+QuickGELU, +residual (+LayerNorm partial sums), fused LayerNorm
    (+QuickGELU), 16-bit and fp32 output."""
    lib = ops._capi.load()
    if os.environ.get("LECLIP_TEST_GEMM_FAMILY"):
        pytest.skip("compares the families under its own overrides")
    g = torch.Generator(device="cpu").manual_seed(11)
    for (M, N, K) in ((50432, 768, 768), (30000, 2304, 768), (20000, 768, 3072)):
        ms = 197 * 3 + 5
        assert lib.leclip_gemm_kernel_name(M, N, K, ops.dtype_code(dt)) in (b"gemm_tn_384x256x32_pp", b"gemm_tn_256x256x64_pp")
        assert lib.leclip_gemm_kernel_name(ms, N, K, ops.dtype_code(dt)) == b"gemm_tn_128x128x64"
        a = torch.randn(M, K, generator=g).to(dt).to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(DEV)
        bias = torch.randn(N, generator=g).to(DEV)
        res = torch.randn(M, N, generator=g).to(dt).to(DEV)
        stats = torch.stack([torch.randn(M, generator=g) * 0.1, torch.rand(M, generator=g) + 0.5], dim=1).contiguous().to(DEV)
        colsum = torch.randn(N, generator=g).to(DEV)
        lo = slice(M - ms, M)       # the LAST rows of the big call (ragged edge tile of the 256 kernel) vs a small call on them
        sa, sres, sstats = a[lo].contiguous(), res[lo].contiguous(), stats[lo].contiguous()
        cases = [
            (dict(), dict()),
            (dict(bias=bias, act=ops.ACT_QUICKGELU), dict(bias=bias, act=ops.ACT_QUICKGELU)),
            (dict(bias=bias, residual=res), dict(bias=bias, residual=sres)),
            (dict(bias=bias, out_dtype=torch.float32), dict(bias=bias, out_dtype=torch.float32)),
        ]
        for fam in (384, 256):      # the big call through each persistent family (round 5: leclip_set_gemm_family), the small one through 128x128
            prev = ops.set_gemm_family(fam)
            try:
                big = [ops.gemm(a, w, **kb) for kb, _ in cases]
                pb = torch.zeros(N // 64, M, 2, device=DEV)
                big.append(ops.gemm_ln(a, w, bias, residual=res, stats_out=pb))
                big += [ops.gemm_ln(a, w, bias, ln_stats=stats, ln_colsum=colsum, act=act) for act in (ops.ACT_NONE, ops.ACT_QUICKGELU)]
            finally:
                ops.set_gemm_family(prev)
            small = [ops.gemm(sa, w, **ks) for _, ks in cases]
            ps = torch.zeros(N // 64, ms, 2, device=DEV)
            small.append(ops.gemm_ln(sa, w, bias, residual=sres, stats_out=ps))
            small += [ops.gemm_ln(sa, w, bias, ln_stats=sstats, ln_colsum=colsum, act=act) for act in (ops.ACT_NONE, ops.ACT_QUICKGELU)]
            for yb, ys in zip(big, small):
                assert torch.equal(yb[lo], ys), fam
            assert torch.equal(pb[:, lo], ps), fam
            del big, small
        del a, w, res


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gemm_res_stats_merges_in_the_producer(ops, dt):
    """leclip_gemm_res_stats_fwd (round 5): the residual GEMM finishes the (mean, rstd) of its output rows - on the 384 x 256 kernel inside the
    launch (the 384-row block's last-arriving workgroup merges the three N-tiles' partials: write-through stores, a returning ticket, sc1 loads),
    elsewhere by the merge kernel behind the GEMM.  Output, partials and statistics must equal the two-launch path bit for bit: full and ragged M,
    every 384-row block, repeated launches on the same ticket words (they must come back to zero), other work in between (the hand-off must
    not depend on what the caches hold), and a small M that takes the 128 x 128 kernel + merge launch."""
    g = torch.Generator(device="cpu").manual_seed(5)
    for (M, N, K) in ((50432, 768, 768), (25216, 768, 3072), (25216 - 300, 768, 768), (4000, 1024, 1024), (1576, 768, 768)):
        a = torch.randn(M, K, generator=g).to(dt).to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(DEV)
        bias = torch.randn(N, generator=g).to(DEV)
        res = (torch.randn(M, N, generator=g) + 3.0).to(dt).to(DEV)
        part_ref = torch.zeros(N // 64, M, 2, device=DEV)
        y_ref = ops.gemm_ln(a, w, bias, residual=res, stats_out=part_ref)
        st_ref = ops.ln_stats_finalize(part_ref, N)
        tickets = torch.zeros((M + 383) // 384, dtype=torch.int32, device=DEV)
        junk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
        for rep in range(6):
            part = torch.full((N // 64, M, 2), float("nan"), device=DEV)
            st = torch.full((M, 2), float("nan"), device=DEV)
            if rep % 2:
                junk.fill_(rep)                              # other traffic between the launches
                _ = st_ref.sum(); _ = part_ref[:, ::7].sum()  # and the reference lines warm in the caches
            y = ops.gemm_res_stats(a, w, bias, res, part, st, tickets)
            assert torch.equal(y, y_ref) and torch.equal(part, part_ref), (M, K, rep)
            assert torch.equal(st, st_ref), (M, K, rep, int((st != st_ref).sum()))
            assert int(tickets.abs().sum()) == 0, "ticket counters must return to zero"
        st = torch.full((M, 2), float("nan"), device=DEV)
        ops.gemm_res_stats(a, w, bias, res, torch.zeros_like(part_ref), st, None)     # no tickets: GEMM + merge launch
        assert torch.equal(st, st_ref)
        del a, w, res


@pytest.mark.parametrize("family", [256, 384], indirect=True)
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_walk_order_never_changes_results(ops, dt, family):
    """leclip_set_walk_order (ABI 8) only picks which rows a workgroup takes first: the reversed tile walk of both persistent GEMM families - ragged M,
    even and odd K-tile counts (the 256x256 kernel pipelines across tiles only for even ones; its edge tile then runs FIRST), residual + partials and
    fused-LayerNorm flavours - and of the many-heads attention kernel (T = 197, whose default is descending) must give the same bits as the ascending one."""
    g = torch.Generator(device="cpu").manual_seed(23)
    for (M, N, K) in ((50432 - 150, 768, 768), (40000 + 77, 768, 320)):
        a = torch.randn(M, K, generator=g).to(dt).to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dt).to(DEV)
        bias = torch.randn(N, generator=g).to(DEV)
        res = torch.randn(M, N, generator=g).to(dt).to(DEV)
        stats = torch.stack([torch.randn(M, generator=g) * 0.1, torch.rand(M, generator=g) + 0.5], dim=1).contiguous().to(DEV)
        colsum = torch.randn(N, generator=g).to(DEV)
        outs = {}
        for order in (-1, 0, 1):
            prev = ops.set_walk_order(order)
            try:
                part = torch.zeros(N // 64, M, 2, device=DEV)
                outs[order] = (ops.gemm(a, w, bias), ops.gemm_ln(a, w, bias, residual=res, stats_out=part), part,
                               ops.gemm_ln(a, w, bias, ln_stats=stats, ln_colsum=colsum, act=ops.ACT_QUICKGELU))
            finally:
                ops.set_walk_order(prev)
        for order in (0, 1):
            for x, y in zip(outs[-1], outs[order]):
                assert torch.equal(x, y), (M, K, order)
        del a, w, res, outs
    if family == 256:
        qkv = (torch.randn(90 * 197, 3 * 768, generator=g) * 0.8).to(dt).to(DEV)
        ys = {}
        for order in (-1, 0, 1):
            prev = ops.set_walk_order(order)
            try:
                ys[order] = ops.attention(qkv, 90, 197, 12, False)
            finally:
                ops.set_walk_order(prev)
        assert torch.equal(ys[-1], ys[0]) and torch.equal(ys[-1], ys[1])


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 197, 12, False), (3, 77, 8, True), (2, 17, 2, False), (1, 50, 1, True), (2, 224, 1, False),
                                 (1, 1, 1, True)])
def test_attention(ops, dt, cfg):
    b, t, h, causal = cfg
    d = 64 * h
    qkv = _rand((b * t, 3 * d), 11).to(dt)
    q, k, v = [z.reshape(b, t, h, 64).transpose(1, 2).double() for z in qkv.float().split(d, dim=-1)]
    s = q @ k.transpose(-1, -2) * 0.125
    if causal:
        s = s + torch.triu(torch.full((t, t), float("-inf"), dtype=torch.float64), 1)
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(b * t, d)
    y = ops.attention(qkv.to(DEV), b, t, h, causal)
    np.testing.assert_allclose(y.double().cpu().numpy(), ref.numpy(), atol=_tol(dt, 2e-5, 4e-3, 2.5e-2), rtol=0)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(1, 577, 3, False), (2, 640, 1, False), (1, 257, 2, False), (1, 225, 1, False), (1, 512, 1, False), (1, 300, 2, True),
                                 (1, 385, 1, True), (1, 640, 1, True)])
def test_attention_long(ops, dt, cfg):
    """225 .. 640 tokens run the streaming kernel (attn_stream_kernel: 128-key chunks, online softmax with a lazy reference maximum): whole chunks only (512), a last
    chunk of one key (257, 385), of one tile plus one key (225 -> 97 keys in chunk 1), of 65 keys (577: ViT-L/14@336), the full 640, and
    the causal form, whose chunks are masked per element and end at the query block's diagonal."""
    b, t, h, causal = cfg
    d = 64 * h
    qkv = _rand((b * t, 3 * d), 13).to(dt)
    q, k, v = [z.reshape(b, t, h, 64).transpose(1, 2).double() for z in qkv.float().split(d, dim=-1)]
    s = q @ k.transpose(-1, -2) * 0.125
    if causal:
        s = s + torch.triu(torch.full((t, t), float("-inf"), dtype=torch.float64), 1)
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(b * t, d)
    y = ops.attention(qkv.to(DEV), b, t, h, causal)
    np.testing.assert_allclose(y.double().cpu().numpy(), ref.numpy(), atol=_tol(dt, 2e-5, 4e-3, 2.5e-2), rtol=0)
    assert torch.equal(y, ops.attention(qkv.to(DEV), b, t, h, causal))     # deterministic


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 577, 3, False), (1, 600, 2, True), (1, 257, 2, False), (1, 640, 1, False)])
def test_attention_long_late_maxima(ops, dt, cfg):
    """The streaming kernel keeps a LAZY reference maximum per query (set by the first 128-key chunk; the S accumulators start from
    -m_ref, a later chunk moves it only when it exceeds it by more than 2^8).  Key rows that are multiples of query rows put a query's
    maximum into a LATER chunk: far above the threshold (factor 3 - 4: the rescale branch), below it (0.6: the stale reference is kept),
    in the last tile, in the chunk after the first, and twice for one query."""
    b, t, h, causal = cfg
    d = 64 * h
    x = _rand((b * t, 3 * d), 17).reshape(b, t, 3, h, 64).clone()
    pairs = [(5, 140, 3.0), (40, 300, 3.0), (41, 520, 0.6), (t - 3, t - 2, 3.0), (t - 1, 130, 0.6), (200, 131, 3.0), (200, 400, 4.0),
             (333, t - 1, 3.0), (64, 64, 3.0), (65, 257, 2.0), (130, 129, 1.5)]
    for qi, ki, f in pairs:
        if 0 <= qi < t and 0 <= ki < t:
            x[:, ki, 1] = f * x[:, qi, 0]
    qkv = x.reshape(b * t, 3 * d).to(dt)
    q, k, v = [z.reshape(b, t, h, 64).transpose(1, 2).double() for z in qkv.float().split(d, dim=-1)]
    s = q @ k.transpose(-1, -2) * 0.125
    if causal:
        s = s + torch.triu(torch.full((t, t), float("-inf"), dtype=torch.float64), 1)
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(b * t, d)
    y = ops.attention(qkv.to(DEV), b, t, h, causal)
    assert bool(torch.isfinite(y.float()).all())
    np.testing.assert_allclose(y.double().cpu().numpy(), ref.numpy(), atol=_tol(dt, 2e-5, 4e-3, 2.5e-2), rtol=0)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(86, 197, 12), (43, 224, 24), (22, 193, 48), (300, 197, 12), (86, 200, 12), (86, 201, 12), (86, 208, 12), (86, 209, 12)])
def test_attention_many_heads(ops, dt, cfg):
    """>= 1024 (batch, head) pairs with 193..224 tokens run the persistent pipelined kernel (attn_heads_kernel): head counts
    that are not a multiple of the CU count leave workgroups with different numbers of heads.  T = 193..200 / 201..208 / 209..224 are the kernel's
    three instances (1 / 2 / all 4 of the last key tile's 8-key groups computed): each boundary is a case, and each must give the bits of the
    one-workgroup-per-head kernel, which computes and masks every group."""
    b, t, h = cfg
    d = 64 * h
    qkv = _rand((b * t, 3 * d), 31).to(dt)
    y = ops.attention(qkv.to(DEV), b, t, h, False)
    q, k, v = [z.reshape(b, t, h, 64).transpose(1, 2) for z in qkv.to(DEV).float().split(d, dim=-1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(b * t, d)   # fp32 torch on the same device
    err = float((y.float() - ref).abs().max())
    assert err <= _tol(dt, 2e-5, 4e-3, 2.5e-2), err
    assert torch.equal(y, ops.attention(qkv.to(DEV), b, t, h, False))     # deterministic
    nb = max(1, 1023 // h)                                                 # < 1024 pairs: attn_rows_kernel
    assert nb < b and torch.equal(y[:nb * t], ops.attention(qkv[:nb * t].to(DEV), nb, t, h, False))
    if cfg in ((86, 197, 12), (86, 201, 12)):                              # the same kernel instances under the causal mask (no model runs them so: kept correct anyway)
        yc = ops.attention(qkv.to(DEV), b, t, h, True)
        mask = torch.full((t, t), float("-inf"), device=DEV).triu(1)
        refc = (torch.softmax(q @ k.transpose(-1, -2) * 0.125 + mask, -1) @ v).transpose(1, 2).reshape(b * t, d)
        assert float((yc.float() - refc).abs().max()) <= _tol(dt, 2e-5, 4e-3, 2.5e-2)
        assert torch.equal(yc[:nb * t], ops.attention(qkv[:nb * t].to(DEV), nb, t, h, True))


@pytest.mark.parametrize("dt", DTYPES)
def test_gather_ln_proj_and_logits(ops, dt):
    from oracle import clip_oracle as co
    x = _rand((40, 768), 12, 2.0).to(dt)
    idx = torch.tensor([0, 39, 7, 7, 13], dtype=torch.int64)
    g, b, proj = _rand((768,), 13) * 0.1 + 1, _rand((768,), 14) * 0.1, _rand((768, 512), 15, 768 ** -0.5).to(dt)
    ref = co.layer_norm(x.float()[idx], g, b) @ proj.float()
    y = ops.gather_ln_proj(x.to(DEV), idx.to(DEV), g.to(DEV), b.to(DEV), proj.to(DEV))
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), atol=3e-5 * float(ref.abs().max()) + 1e-5, rtol=0)
    fi, ft = _rand((9, 512), 16), _rand((80, 512), 17)
    lg = ops.l2norm_logits(fi.to(DEV), ft.to(DEV), 4.0)
    np.testing.assert_allclose(lg.cpu().numpy(), co.cosine_logits(fi, ft, 4.0).numpy(), atol=2e-6, rtol=0)


def test_embedding_kernels(ops, golden_dir):
    t = np.load(os.path.join(golden_dir, "tokens_coco80.npz"))
    toks = torch.from_numpy(t["tokens_ctx16"])
    table, pos = _rand((49408, 128), 18), _rand((77, 128), 19, 0.01)
    x = ops.embed_tokens(toks.to(DEV), table.to(DEV), pos.to(DEV), torch.float32)
    assert torch.equal(x.cpu(), table[toks] + pos)
    eot, flat = ops.eot_index(toks.to(DEV))
    assert torch.equal(eot.cpu(), toks.argmax(-1)) and torch.equal(flat.cpu(), torch.arange(80) * 77 + toks.argmax(-1))
    emb = table[toks]
    ctx = _rand((16, 128), 20, 0.02)
    ref = torch.cat([emb[:, :1], ctx.expand(80, -1, -1), emb[:, 17:]], 1)
    out = ops.prompt_assemble(emb[:, :1].contiguous().to(DEV), ctx.to(DEV), emb[:, 17:].contiguous().to(DEV), None, torch.float32)
    assert torch.equal(out.cpu(), ref)
    ctx_c = _rand((80, 16, 128), 21, 0.02)
    out = ops.prompt_assemble(emb[:, :1].contiguous().to(DEV), ctx_c.to(DEV), emb[:, 17:].contiguous().to(DEV), pos.to(DEV), torch.float32)
    assert torch.equal(out.cpu(), torch.cat([emb[:, :1], ctx_c, emb[:, 17:]], 1) + pos)
    assert torch.equal(ops.add_pos(ref.to(DEV), pos.to(DEV), torch.float32).cpu(), ref + pos)
    # ties: argmax returns the first maximum, like torch
    tie = torch.tensor([[5, 9, 9, 1], [0, 0, 0, 0]], dtype=torch.int64)
    assert torch.equal(ops.eot_index(tie.to(DEV))[0].cpu(), tie.argmax(-1))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", [(32, 8, 128), (224, 16, 768), (28, 14, 128)])
def test_patch_embed(ops, dt, geom):
    from oracle import clip_oracle as co
    r, p, width = geom
    img = _rand((3, 3, r, r), 22)
    w = _rand((width, 3, p, p), 23, (3 * p * p) ** -0.5).to(dt)
    cls, pos = _rand((width,), 24), _rand(((r // p) ** 2 + 1, width), 25)
    sd = {"visual.conv1.weight": w.float(), "visual.class_embedding": cls, "visual.positional_embedding": pos}
    ref = co.patch_embed(img.to(dt).float(), sd)
    k = 3 * p * p
    al = 32 if dt == torch.float32 else 64
    kp = (k + al - 1) // al * al
    wp = torch.zeros((width, kp), dtype=dt)
    wp[:, :k] = w.reshape(width, k)
    x = ops.patch_embed(img.to(DEV), wp.to(DEV), cls.to(DEV), pos.to(DEV), p, dt)
    np.testing.assert_allclose(x.float().cpu().numpy(), ref.numpy(), atol=_tol(dt, 1e-5, 6e-3, 5e-2), rtol=0)
    # the same pixel values handed over in the compute dtype (what bench.py does): the extraction kernels' 16-bit-input forms - 8-pixel chunks
    # for 8 | P, pixel pairs for patch 14 (round 5) - must give the same bits as their fp32-input forms
    x16 = ops.patch_embed(img.to(dt).to(DEV), wp.to(DEV), cls.to(DEV), pos.to(DEV), p, dt)
    assert torch.equal(x16, x)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("batch", [96, 85])
def test_patch_embed_im2col_free(ops, dt, batch):
    """Round 4: images handed over in the compute dtype take the im2col-free patch GEMM (the 256 x 256 kernel gathers its A tiles from the
    NCHW image by LDS-DMA, no patch matrix: csrc/gemm_mfma256.hip PP<.., IM2COL>) once the batch fills that kernel (B >= 84 at 224 x 224).
    The same pixel values handed over as fp32 take the older path (patch-extraction kernel, then the same GEMM on a patch matrix): both
    must produce the residual stream and its LayerNorm partials BIT FOR BIT, and match ln_pre(patch_embed) of the oracle.  B = 85: a
    last tile row that is partly past M (clamped gather rows, clipped stores)."""
    from oracle import clip_oracle as co
    r, p, width = 224, 16, 768
    img16 = _rand((batch, 3, r, r), 31).to(dt)
    w = _rand((width, 3, p, p), 32, (3 * p * p) ** -0.5).to(dt)
    cls, pos = _rand((width,), 33), _rand(((r // p) ** 2 + 1, width), 34)
    gamma, beta = 1 + 0.1 * _rand((width,), 35), 0.1 * _rand((width,), 36)
    wp = w.reshape(width, 3 * p * p).contiguous().to(DEV)
    args = (wp, cls.to(DEV), pos.to(DEV), gamma.to(DEV), beta.to(DEV), p, dt)
    t = (r // p) ** 2 + 1
    st_a = torch.zeros((width // 64, batch * t, 2), dtype=torch.float32, device=DEV)
    st_b = torch.zeros_like(st_a)
    xa = ops.patch_embed_ln(img16.to(DEV), *args, stats_out=st_a)                    # compute-dtype image: im2col-free
    xb = ops.patch_embed_ln(img16.float().to(DEV), *args, stats_out=st_b)            # same values as fp32: patch matrix + GEMM
    assert xa.dtype == dt and torch.equal(xa, xb) and torch.equal(st_a, st_b)
    sd = {"visual.conv1.weight": w.float(), "visual.class_embedding": cls, "visual.positional_embedding": pos}
    sel = torch.tensor([0, 1, batch // 2, batch - 1])
    ref = co.layer_norm(co.patch_embed(img16[sel].float(), sd), gamma, beta)
    np.testing.assert_allclose(xa[sel].float().cpu().numpy(), ref.numpy(), atol=_tol(dt, 1e-5, 1.5e-2, 1e-1), rtol=0)


# --------------------------------------------------------------------------------- towers against the oracle / golden
def _build(arch, seed, dist, dtype):
    from leclip_amd.clip import build_model, convert_weights
    m = build_model(synth.make_state_dict(arch, seed=seed, dist=dist))
    m.float()
    if dtype != torch.float32:
        convert_weights(m, dtype)
    return m.to(DEV)


@pytest.mark.parametrize("dt", DTYPES)
def test_tiny_per_stage(ops, golden_dir, dt):
    g = np.load(os.path.join(golden_dir, "tiny_stages.npz"))
    m = _build(synth.TINY, 1, "cond", dt)
    taps = {}
    feat = m.visual(torch.from_numpy(g["images"]).to(DEV), taps)
    tol = _tol(dt, 2e-4, 3e-2, 2.5e-1)
    np.testing.assert_allclose(taps["ln_pre"].cpu().numpy().reshape(g["v.ln_pre"].shape), g["v.ln_pre"], atol=tol, rtol=tol)
    for i in range(synth.TINY.vision_layers):
        for k in ("ln_1", "ln_2", "gelu", "out"):
            got = taps[f"block{i}.{k}"].cpu().numpy().reshape(g[f"v.block{i}.{k}"].shape)
            np.testing.assert_allclose(got, g[f"v.block{i}.{k}"], atol=tol, rtol=tol, err_msg=f"block{i}.{k}")
    np.testing.assert_allclose(feat.cpu().numpy(), g["v.feat"], atol=tol, rtol=tol)
    toks = torch.from_numpy(g["tokens"]).to(DEV)
    taps = {}
    tf = m.text_engine(torch.device(DEV)).encode_tokens(toks, taps=taps)
    for i in range(synth.TINY.transformer_layers):
        for k in ("ln_1", "ln_2", "gelu", "out"):
            got = taps[f"block{i}.{k}"].cpu().numpy().reshape(g[f"t.block{i}.{k}"].shape)
            np.testing.assert_allclose(got, g[f"t.block{i}.{k}"], atol=tol, rtol=tol, err_msg=f"text block{i}.{k}")
    np.testing.assert_allclose(tf.cpu().numpy(), g["t.feat"], atol=tol, rtol=tol)
    lpi, lpt = m(torch.from_numpy(g["images"]).to(DEV), toks)
    ltol = _tol(dt, 1e-3, 2e-2, 1.5e-1)  # scale exp(logit_scale) = 14.3
    np.testing.assert_allclose(lpi.cpu().numpy(), g["logits_per_image"], atol=ltol, rtol=0)
    np.testing.assert_allclose(lpt.cpu().numpy(), g["logits_per_text"], atol=ltol, rtol=0)


def _margin_ok(ref_logits, band):
    s = np.sort(ref_logits, axis=1)
    return (s[:, -1] - s[:, -2]) > 2 * band


@pytest.mark.parametrize("dist", ["cond", "default"])
@pytest.mark.parametrize("dt", DTYPES)
def test_vitb16_cfg1_golden(ops, golden_dir, dt, dist):
    """BASELINE config 1 (ViT-B/16, 80 prompts, B=8) against the reference forward's golden outputs."""
    g = np.load(os.path.join(golden_dir, "vitb16_cfg1.npz"))
    t = np.load(os.path.join(golden_dir, "tokens_coco80.npz"))
    m = _build(synth.VIT_B16, 0, dist, dt)
    img = torch.from_numpy(synth.make_images(8, 224, seed=1234)).to(DEV)
    toks = torch.from_numpy(t["tokens_photo"]).to(DEV)
    lpi, _ = m(img, toks)
    lpi = lpi.cpu().numpy()
    ref = g[dist + ".logits_clip"]
    band = _tol(dt, 1e-3, 1.5e-2, 1.2e-1)  # scale 14.29: cosine error x 14.29
    np.testing.assert_allclose(lpi, ref, atol=band, rtol=0)
    if dt == torch.float32:
        assert np.array_equal(np.argsort(-lpi, axis=1, kind="stable")[:, :5], g[dist + ".top5_clip"])
        fi = m.encode_image(img).cpu().numpy()
        np.testing.assert_allclose(fi, g[dist + ".image_features"], atol=1e-3, rtol=1e-3)
    else:
        ok = _margin_ok(ref, band)
        assert np.array_equal(lpi.argmax(1)[ok], ref.argmax(1)[ok])


# constant label-index bands (VERDICT r3 weak 2 / task 7) = the a-priori logit tolerances of this file at scale 4.0 (fp32: the north star's 1e-3; fp16 4e-3,
# bf16 4e-2; measured maxima over rounds 2 - 4: 5e-6, 2.5e-3 .. 3.2e-3, 2.8e-2 .. 3.3e-2): two reference logits closer than the band cannot be ordered by
# arithmetic with that error, a disagreement outside it is a wrong result; a uniformly worse kernel cannot widen its own band (bench.py uses the same numbers).
LABEL_BAND = {torch.float32: 1e-3, torch.float16: 4e-3, torch.bfloat16: 4e-2}


@pytest.mark.parametrize("dt", DTYPES)
def test_vitb16_outlier_golden(ops, golden_dir, dt):
    """ViT-B/16, B=8, weights with the statistics of a released checkpoint (synth dist="outlier": three massive-activation channels per
    residual stream, 50 - 90 sigma after ln_pre and +-140 - 240 after the last block, damped LayerNorm gains, non-zero row means)
    against the REFERENCE's forward on the same weights (tests/golden/vitb16_outlier.npz).  Stresses exactly what the benign sets do not:
    the folded LayerNorm's rstd * (x.W'^T - mean * colsum) on rows whose variance three channels own, and 16-bit rounding where it is
    coarse."""
    from leclip_amd.config import get_cfg_default
    from leclip_amd.datasets import coco_object_categories
    from leclip_amd.trainers import CustomCLIP
    g = np.load(os.path.join(golden_dir, "vitb16_outlier.npz"))
    t = np.load(os.path.join(golden_dir, "tokens_coco80.npz"))
    sd = synth.make_state_dict(synth.VIT_B16, seed=0, dist="outlier")
    assert np.array_equal(sd["visual.ln_pre.bias"].numpy()[:16], g["guard.ln_pre_bias_head"])     # generator drift guard
    m = _build(synth.VIT_B16, 0, "outlier", dt)
    img = torch.from_numpy(synth.make_images(8, 224, seed=1234)).to(DEV)
    lpi, _ = m(img, torch.from_numpy(t["tokens_photo"]).to(DEV))
    lpi = lpi.cpu().numpy()
    err = float(np.abs(lpi - g["logits_clip"]).max())
    fi = m.encode_image(img).float().cpu().numpy()
    cc = CustomCLIP(get_cfg_default(), coco_object_categories, m.cpu())
    with torch.no_grad():
        cc.prompt_learner.ctx.copy_(torch.from_numpy(synth.make_ctx(16, 512, seed=0)))
    cc.to(DEV).eval()
    with torch.no_grad():
        lcc = cc(img, if_test=True)[0].float().cpu().numpy()
    err_cc = float(np.abs(lcc - g["logits_custom_ctx16"]).max())
    print(f"outlier weights [{dt}]: max|dlogit| CLIP (scale 14.3) {err:.3e}, CustomCLIP (scale 4) {err_cc:.3e}")
    # measured (round 4): fp32 3.8e-5 / 5.5e-6, fp16 2.3e-2 / 2.7e-3, bf16 2.3e-1 / 2.2e-2 - the same error per unit of logit scale as on the benign
    # weight sets; bounds = 2 x measured (fp32: the north star's 1e-3)
    assert err <= _tol(dt, 1e-3, 4.6e-2, 4.6e-1) and err_cc <= _tol(dt, 1e-3, 5.5e-3, 4.5e-2), (err, err_cc)
    if dt == torch.float32:
        assert np.array_equal(np.argsort(-lpi, axis=1, kind="stable")[:, :5], g["top5_clip"])
        assert np.array_equal(np.argsort(-lcc, axis=1, kind="stable")[:, :5], g["top5_custom_ctx16"])
        np.testing.assert_allclose(fi, g["image_features"], atol=1e-3, rtol=1e-3)
    else:
        ok = _margin_ok(g["logits_custom_ctx16"], LABEL_BAND[dt] / 2)
        assert np.array_equal(lcc.argmax(1)[ok], g["logits_custom_ctx16"].argmax(1)[ok])


def _cfg4_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "vitb16_cfg4_logits.npz"))
    labels = np.unpackbits(g["labels"], axis=1)[:, :int(g["n_classes"])].astype(np.int64)
    return g["logits"], labels, float(g["mAP_reference"])


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16, torch.float32])
def test_cfg4_logits_against_the_reference_at_size(ops, golden_dir, dt):
    """BASELINE configs[3] at the survey's N = 2 048, pinned to the REFERENCE itself: tests/golden/vitb16_cfg4_logits.npz
    holds the logits of the reference's model.py on the eight ranks' images (oracle/make_golden.py cfg4_goldens), the labels drawn from
    them and the reference mAP().  Per dtype: max |logit difference|, label-index agreement outside the CONSTANT band, |mAP - reference|.
    This is what settles the bf16 clause: at N = 2 048 the sampling noise of 192-image samples is gone."""
    from leclip_amd.config import get_cfg_default
    from leclip_amd.datasets import coco_object_categories
    from leclip_amd.evaluation import mAP
    from leclip_amd.trainers import CustomCLIP
    ref, labels, m_ref = _cfg4_fixture(golden_dir)
    assert abs(mAP(labels, ref) - m_ref) < 1e-9          # our mAP() on the fixture's rows == the reference's mAP() value
    m = _build(synth.VIT_B16, 0, "cond", dt).cpu()
    cc = CustomCLIP(get_cfg_default(), coco_object_categories, m)
    with torch.no_grad():
        cc.prompt_learner.ctx.copy_(torch.from_numpy(synth.make_ctx(16, 512, seed=0)))
    cc.to(DEV).eval()
    n = 2048                                            # every dtype on the whole set (VERDICT r4 weak a: the fp32 leg ran a quarter of it; it costs seconds)
    hip = []
    with torch.no_grad():
        for r in range(n // 256):
            img = torch.from_numpy(synth.make_images(256, 224, seed=1234, start=256 * r)).to(DEV)
            hip.append(cc(img, if_test=True)[0].float().cpu().numpy())
    hip = np.concatenate(hip)
    ref = ref[:n]
    err = float(np.abs(hip - ref).max())
    r1, h1 = ref.argmax(1), hip.argmax(1)
    dis = np.nonzero(r1 != h1)[0]
    worst = max([float(ref[i, r1[i]] - ref[i, h1[i]]) for i in dis], default=0.0)     # the oracle's margin to the label the HIP path picked
    line = f"cfg4 [{dt}] N={n}: max|dlogit| {err:.3e}, top-1 agreement {1 - len(dis) / n:.4f} ({len(dis)} differ, worst reference margin {worst:.2e})"
    if n == 2048:
        m_hip = mAP(labels, hip)
        line += f", mAP reference {m_ref:.3f} hip {m_hip:.3f} (delta {m_hip - m_ref:+.3f})"
    print(line)
    assert err <= _tol(dt, 1e-3, 4e-3, 4e-2), err
    assert worst <= LABEL_BAND[dt], (worst, LABEL_BAND[dt])
    if dt == torch.float32:
        assert len(dis) == 0
    elif dt == torch.float16:
        assert abs(m_hip - m_ref) <= 0.2                 # the north star's clause, at size, against the reference
    else:
        # bf16: measured -0.254 at N = 2 048 against the reference (round 4; max |dlogit| 3.2e-2, 56 near-tie flips): it does NOT meet the
        # +-0.2 clause, with the sampling noise of the 192-image samples gone - DESIGN.md section 3: bf16 is the rate-only companion dtype, the
        # clause is claimed for fp16 (the reference's own GPU precision).  The bound here only guards the kernels against a regression.
        assert abs(m_hip - m_ref) <= 0.4


@pytest.mark.parametrize("dt", DTYPES)
def test_custom_clip_golden(ops, golden_dir, dt):
    """CustomCLIP (learnable 16-token context, x4.0 cosine logits), image branch and caption-as-image branch."""
    from leclip_amd.config import get_cfg_default
    from leclip_amd.datasets import coco_object_categories
    from leclip_amd.trainers import CustomCLIP
    g = np.load(os.path.join(golden_dir, "vitb16_cfg1.npz"))
    t = np.load(os.path.join(golden_dir, "tokens_coco80.npz"))
    m = _build(synth.VIT_B16, 0, "cond", dt).cpu()
    cc = CustomCLIP(get_cfg_default(), coco_object_categories, m)
    with torch.no_grad():
        cc.prompt_learner.ctx.copy_(torch.from_numpy(synth.make_ctx(16, 512, seed=0)))
    cc.to(DEV).eval()
    assert np.array_equal(cc.tokenized_prompts.numpy(), t["tokens_ctx16"])
    img = torch.from_numpy(synth.make_images(8, 224, seed=1234)).to(DEV)
    with torch.no_grad():
        logits, a, b, c = cc(img, if_test=True)
        lcap = cc(None, torch.from_numpy(t["tokens_photo"][:6]).to(DEV))[0]
        prompts = cc.prompt_learner()[0]
    assert a is None and b is None and c is None and logits.shape == (8, 80)
    band = _tol(dt, 1e-3, 4e-3, 3.5e-2)
    ref = g["cond.logits_custom_ctx16"]
    np.testing.assert_allclose(logits.cpu().numpy(), ref, atol=band, rtol=0)
    np.testing.assert_allclose(lcap.cpu().numpy(), g["cond.logits_custom_captions"], atol=band, rtol=0)
    if dt == torch.float32:
        assert np.array_equal(np.argsort(-logits.cpu().numpy(), axis=1, kind="stable")[:, :5], g["cond.top5_custom_ctx16"])
    else:
        ok = _margin_ok(ref, band)
        assert np.array_equal(logits.cpu().numpy().argmax(1)[ok], ref.argmax(1)[ok])
    # PromptLearner.forward()[0] is exactly cat(prefix, ctx, suffix)
    sd_tab = m.token_embedding.weight.detach().float().cpu()
    emb = sd_tab[torch.from_numpy(t["tokens_ctx16"])]
    want = torch.cat([emb[:, :1], cc.prompt_learner.ctx.detach().cpu().expand(80, -1, -1), emb[:, 17:]], 1)
    assert torch.equal(prompts.cpu(), want)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_full_batch_properties(ops, dt):
    """BASELINE config 2 size (B=256).  Size-independent properties: run-to-run determinism (bit for bit); SHARD
    INVARIANCE, bit for bit - an image's logits do not depend on which shard of the batch it is scored in (cfg 4: the
    all-gathered logits of 2 / 4 ranks' shards equal the single-GPU run: every GEMM of a call runs ONE kernel family
    whose per-row arithmetic is independent of the row's position, attention is per (image, head), LayerNorm statistics
    are per row with a fixed summation order); BATCH INVARIANCE - a single image (M = 197 rows: the 128x128 GEMM
    family, the non-pipelined attention kernel) gives the same bits as that image inside the batch of 256, because both
    GEMM families accumulate with the same MFMA in the same K order; and the cosine bound |logit| <= 4."""
    from leclip_amd.config import get_cfg_default
    from leclip_amd.datasets import coco_object_categories
    from leclip_amd.trainers import CustomCLIP
    m = _build(synth.VIT_B16, 0, "cond", dt).cpu()
    cc = CustomCLIP(get_cfg_default(), coco_object_categories, m).to(DEV).eval()
    img = torch.from_numpy(synth.make_images(256, 224, seed=77)).to(DEV)
    with torch.no_grad():
        full = cc(img, if_test=True)[0].clone()
        again = cc(img, if_test=True)[0].clone()
        lo = cc(img[:128].contiguous(), if_test=True)[0].clone()
        hi = cc(img[128:].contiguous(), if_test=True)[0].clone()
        ragged = torch.cat([cc(img[a:b].contiguous(), if_test=True)[0] for a, b in ((0, 100), (100, 187), (187, 256))])
        one = cc(img[200:201].contiguous(), if_test=True)[0].clone()
        # every token's feature (the local branch, N4): the batch as stream parts == one image at a time through the small-batch kernels
        dense = cc.image_encoder.dense_features(img)
        dense_1 = torch.cat([cc.image_encoder.dense_features(img[i:i + 1].contiguous()) for i in (0, 127, 128, 255)])
    assert dense.shape == (256, 197, 512) and torch.equal(dense[[0, 127, 128, 255]], dense_1)
    del dense, dense_1
    # forward_beside: the callable's work is enqueued between the fork and the join of the stream parts (large batch), or simply after the tower
    # (small batch, empty shard); the features are forward()'s bits and the callable's result comes back with them
    with torch.no_grad():
        vis = cc.image_encoder
        probe = torch.arange(8, device=DEV, dtype=torch.float32)
        for sl in (slice(0, 256), slice(0, 3), slice(0, 0)):
            part = img[sl].contiguous()
            feats, res = vis.forward_beside(part, lambda: probe * 2)
            assert torch.equal(feats, vis(part)) and torch.equal(res, probe * 2) and vis.engine(DEV).beside_result is None
    assert torch.isfinite(full).all() and torch.equal(full, again)
    assert torch.equal(full[:128], lo) and torch.equal(full[128:], hi)      # 2 equal shards == unsharded, bit for bit
    assert torch.equal(full, ragged)                                        # ragged shards (100 / 87 / 69 images) too
    assert torch.equal(full[200:201], one)                                  # B = 1 == the same image inside B = 256
    assert float(full.abs().max()) <= 4.0 + 1e-4   # |cos| <= 1 scaled by 4
    # The engine runs B >= 128 as two parts on two HIP streams (kernel tails of one part under the other's kernels): one part,
    # three parts and an uneven split give the same bits, features included.
    eng = cc.image_encoder.engine(img.device)
    assert eng.streams == 2 and eng._parts(img) == [(0, 128), (128, 256)]
    with torch.no_grad():
        feats2 = cc.image_encoder(img).clone()
        try:
            eng.streams = 1
            assert eng._parts(img) is None
            single = cc(img, if_test=True)[0].clone()
            feats1 = cc.image_encoder(img).clone()
            eng.streams = 3
            three = cc(img, if_test=True)[0].clone()
            eng.split_sizes = [37, 219]
            uneven = cc(img, if_test=True)[0].clone()
        finally:
            eng.streams, eng.split_sizes = 2, None
    torch.cuda.synchronize()
    assert torch.equal(full, single) and torch.equal(full, three) and torch.equal(full, uneven) and torch.equal(feats1, feats2)
    # The last block computes only what the class token needs (run_blocks cls_last): same bits as the block computed for every token.
    assert eng.cls_last_block
    with torch.no_grad():
        try:
            eng.cls_last_block = False
            whole = cc(img, if_test=True)[0].clone()
            feats_whole = cc.image_encoder(img).clone()
        finally:
            eng.cls_last_block = True
    assert torch.equal(full, whole) and torch.equal(feats2, feats_whole)
    # odd batch sizes through the same rules (129 -> parts of 65 + 64; 65 and 3 -> one part): still the rows of the full batch
    with torch.no_grad():
        for n in (129, 65, 3):
            assert torch.equal(cc(img[256 - n:].contiguous(), if_test=True)[0], full[256 - n:])


@pytest.mark.parametrize("dt,tol,claim", [(torch.float16, 0.2, "north-star clause"), (torch.bfloat16, 0.6, "regression bound only (the clause is settled at N = 2048 by test_cfg4_logits_against_the_reference_at_size: -0.25, not met)")])
def test_map_against_oracle(ops, dt, tol, claim):
    """mAP over 80 labels of the HIP logits vs the fp32 CPU oracle's on the same 256 images, labels drawn from the
    oracle logits.
    fp16 - THE ACCURACY CLAIM: within +-0.2 (north star), the reference's own GPU precision (model.py:470) and the headline
    dtype of bench.py (measured 0.004 .. 0.03).
    bf16 - NO CLAIM: the companion dtype does not meet the +-0.2 clause on every sample (measured 0.17 .. 0.43) and is not
    reported as meeting it (bench.py prints its own gate line for it); 0.6 here only guards against a
    regression of the kernels.  profiles/r02_lowprec_error_budget.py shows the gap is the 8-bit mantissa of the ACTIVATIONS
    themselves (any bf16 run of the reference would carry it; an fp32 residual stream recovers a third of it)."""
    from leclip_amd.config import get_cfg_default
    from leclip_amd.datasets import coco_object_categories
    from leclip_amd.evaluation import mAP
    from leclip_amd.trainers import CustomCLIP
    from oracle import clip_oracle as co
    sd = synth.make_state_dict(synth.VIT_B16, seed=0, dist="cond")
    m = _build(synth.VIT_B16, 0, "cond", dt).cpu()
    cc = CustomCLIP(get_cfg_default(), coco_object_categories, m)
    ctx = torch.from_numpy(synth.make_ctx(16, 512, seed=0))
    with torch.no_grad():
        cc.prompt_learner.ctx.copy_(ctx)
    cc.to(DEV).eval()
    n = 256
    img = torch.from_numpy(synth.make_images(n, 224, seed=4321))
    toks = cc.tokenized_prompts.cpu()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        prefix, suffix = co.prompt_buffers(toks, sd, 16)
        txt = co.text_encoder(co.prompt_learner_forward(ctx, prefix, suffix), toks, sd)
        ref = torch.cat([co.cosine_logits(co.encode_image(img[i:i + 32], sd), txt, 4.0) for i in range(0, n, 32)]).numpy()
        hip = cc(img.to(DEV), if_test=True)[0].float().cpu().numpy()
    labels = synth.make_labels_from_logits(ref, seed=7, pos_frac=0.1, noise=0.5)
    m_ref, m_hip = mAP(labels, ref), mAP(labels, hip)
    print(f"mAP oracle {m_ref:.3f} hip[{dt}] {m_hip:.3f} max|dlogit| {np.abs(ref - hip).max():.3e}  ({claim}: <= {tol})")
    assert m_ref > 30 and abs(m_ref - m_hip) <= tol, claim


@pytest.mark.parametrize("dt", DTYPES)
def test_vit_l14_336_shapes_against_oracle(ops, golden_dir, dt):
    """BASELINE config 5 geometry (ViT-L/14@336: 577 tokens, width 1024, 16 heads, patch 14, text width 768) at reduced
    depth (2 + 2 blocks) against the CPU oracle: exercises the streaming attention kernel, the non-8-multiple patch
    path and the K-padded patch GEMM."""
    import dataclasses
    from oracle import clip_oracle as co
    arch = dataclasses.replace(synth.VIT_L14_336, vision_layers=2, transformer_layers=2)
    sd = synth.make_state_dict(arch, seed=3, dist="cond")
    m = _build(arch, 3, "cond", dt)
    img = torch.from_numpy(synth.make_images(2, 336, seed=9))
    t = np.load(os.path.join(golden_dir, "tokens_coco80.npz"))
    toks = torch.from_numpy(t["tokens_photo"][:7])
    with torch.no_grad():
        ref = co.clip_forward(img, toks, sd).numpy()
        lpi, _ = m(img.to(DEV), toks.to(DEV))
    band = _tol(dt, 1e-3, 2e-2, 1.5e-1)   # scale exp(logit_scale) = 14.3
    np.testing.assert_allclose(lpi.cpu().numpy(), ref, atol=band, rtol=0)
    if dt == torch.float32:
        assert np.array_equal(lpi.cpu().numpy().argmax(1), ref.argmax(1))


def test_cfg5_vit_l14_336_full_depth(ops, golden_dir):
    """BASELINE configs[4] at FULL depth (24 + 12 blocks, 577 tokens, width 1024 / 768) in fp16 - the precision that config
    names - on 8 images x 80 prompts against the fp32 CPU oracle: streaming attention, patch-14 embedding, the large-K
    GEMMs and the fused tail at width 1024 / embed 768."""
    from oracle import clip_oracle as co
    arch = synth.VIT_L14_336
    sd = synth.make_state_dict(arch, seed=3, dist="cond")
    m = _build(arch, 3, "cond", torch.float16)
    img = torch.from_numpy(synth.make_images(8, 336, seed=9))
    t = np.load(os.path.join(golden_dir, "tokens_coco80.npz"))
    toks = torch.from_numpy(t["tokens_photo"])
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        ref = co.clip_forward(img, toks, sd).numpy()
        lpi, _ = m(img.to(DEV), toks.to(DEV))
    lpi = lpi.cpu().numpy()
    err = float(np.abs(lpi - ref).max())
    print(f"ViT-L/14@336 full depth fp16: max|dlogit| {err:.3e} at scale {float(sd['logit_scale'].exp()):.1f}, top1 agree "
          f"{(lpi.argmax(1) == ref.argmax(1)).mean():.3f}")
    assert lpi.shape == (8, 80) and err <= 4e-2          # cosine error <= 2.8e-3 at scale 14.3
    ok = _margin_ok(ref, 4e-2)
    assert np.array_equal(lpi.argmax(1)[ok], ref.argmax(1)[ok])


def test_local_pool_kernel_against_reference_slice(ops, golden_dir):
    """leclip_local_pool_fwd against the outputs of the reference's own pooling lines (:447-462), both variants."""
    g = np.load(os.path.join(golden_dir, "postprocess.npz"))
    ln, le = g["n4.logits_neg"], g["n4.logits_evidence"]            # [P, B, C]
    p, b, c = ln.shape
    t, cp = p + 1, 128
    sim = np.zeros((b, t, 2 * cp), dtype=np.float32)                 # row 0 of every image = the class token (skipped), padded columns
    sim[:, 1:, :c] = ln.transpose(1, 0, 2)
    sim[:, 1:, cp:cp + c] = le.transpose(1, 0, 2)
    sim[:, 0] = 7.0
    d = torch.from_numpy(sim.reshape(b * t, 2 * cp)).to(DEV)
    got = ops.local_pool(d, b, t, 1, c, -1, 40.0, 4.0).cpu().numpy()
    np.testing.assert_allclose(got, g["n4.logits_local.plain"], atol=2e-6, rtol=2e-6)
    got = ops.local_pool(d, b, t, 1, c, cp, 40.0, 4.0).cpu().numpy()
    np.testing.assert_allclose(got, g["n4.logits_local.evidence"], atol=1e-9, rtol=2e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("use_evidence", [False, True])
def test_dense_clip_local_branch(ops, golden_dir, dt, use_evidence):
    """N4 end to end on the tiny model: DenseCLIP (ViT local branch) global and local logits against the oracle's
    dense_clip_forward; then through trainer.test() with TEST.use_freq, where the local scores take the co-occurrence
    modulation (N3) and the evaluator merges global and local scores with GL_merge_rate."""
    from leclip_amd.config import get_cfg_default
    from leclip_amd.datasets import coco_object_categories
    from leclip_amd.trainers import DenseCLIP
    from oracle import clip_oracle as co
    arch = synth.TINY
    sd = synth.make_state_dict(arch, seed=1, dist="cond")
    m = _build(arch, 1, "cond", dt).cpu()
    cfg = get_cfg_default()
    cfg.merge_from_list(["INPUT.SIZE", "(32, 32)", "TRAINER.Caption.use_evidence", str(use_evidence), "TRAIN.MODEL", "DenseCLIP"])
    dc = DenseCLIP(cfg, coco_object_categories, m)
    w = arch.transformer_width
    ctx, ctx2, ctx3 = [torch.from_numpy(synth.make_ctx(16, w, seed=s, name=n) * 10) for s, n in ((0, "ctx"), (1, "ctx_double"), (2, "ctx_evidence"))]
    with torch.no_grad():
        dc.prompt_learner.ctx.copy_(ctx); dc.prompt_learner.ctx_double.copy_(ctx2); dc.prompt_learner.ctx_evidence.copy_(ctx3)
    dc.to(DEV).eval()
    img = torch.from_numpy(synth.make_images(5, 32, seed=11))
    toks = dc.tokenized_prompts
    prefix, suffix = co.prompt_buffers(toks, sd, 16)
    with torch.no_grad():
        ref_g, ref_l = co.dense_clip_forward(img, sd, ctx, ctx2, ctx3 if use_evidence else None, prefix, suffix, toks, 40.0, 4.0)
        out = dc(img.to(DEV), if_test=True)
    assert len(out) == 5 and out[0].shape == (5, 80) and out[1].shape == (5, 80)
    tol = _tol(dt, 1e-3, 2e-2, 1.5e-1)
    np.testing.assert_allclose(out[0].cpu().numpy(), ref_g.numpy(), atol=tol, rtol=0)
    scale_l = float(ref_l.abs().max())
    np.testing.assert_allclose(out[1].cpu().numpy(), ref_l.numpy(), atol=tol * max(scale_l, 1e-3) * 2, rtol=0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,T,h,causal", [(3, 50, 2, False), (2, 77, 8, True), (4, 197, 12, False), (1, 577, 16, False)])
def test_attention_prefix(ops, dt, B, T, h, causal):
    """leclip_attention_prefix_fwd: the first q_rows query rows of every (batch, head) have the bits of the full call; rows past the
    last computed block are not touched."""
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(B * T, 3 * h * 64, generator=g) * 0.7).to(dt).to(DEV)
    full = ops.attention(qkv, B, T, h, causal)
    for q_rows in (1, 33):
        if q_rows > T:
            continue
        out = torch.full_like(full, 7.0)
        ops.attention(qkv, B, T, h, causal, out=out, q_rows=q_rows)
        got, ref = out.view(B, T, -1), full.view(B, T, -1)
        assert torch.equal(got[:, :q_rows], ref[:, :q_rows])
        blk = 1 if dt == torch.float32 else 32
        end = min(T, (q_rows + blk - 1) // blk * blk)
        assert bool((got[:, end:] == 7.0).all())
    with pytest.raises(Exception):
        ops.attention(qkv, B, T, h, causal, q_rows=T + 1)


def test_empty_batch_and_rejected_inputs(ops):
    """Edge inputs of the image branch: an EMPTY batch (the shard of a rank that got no image) returns [0, C] logits / [0, E] features
    without a launch, as torch modules do; a CPU tensor or a wrong resolution raises instead of being routed anywhere else."""
    from leclip_amd.config import get_cfg_default
    from leclip_amd.datasets import coco_object_categories
    from leclip_amd.trainers import CustomCLIP
    arch = synth.TINY
    m = _build(arch, 0, "cond", torch.float16).cpu()
    cfg = get_cfg_default()
    cfg.INPUT.SIZE = (arch.image_resolution, arch.image_resolution)
    cc = CustomCLIP(cfg, coco_object_categories[:7], m).to(DEV).eval()
    img = torch.from_numpy(synth.make_images(3, arch.image_resolution, seed=1)).to(DEV)
    with torch.no_grad():
        full = cc(img, if_test=True)[0]
        empty = cc(img[:0], if_test=True)[0]
        one = cc(img[2:3], if_test=True)[0]
        feats = cc.model.encode_image(img[:0].half())
        dense = cc.image_encoder.dense_features(img[:0].half())
        with pytest.raises(RuntimeError, match="HIP device"):
            cc(img.cpu(), if_test=True)
        with pytest.raises(ValueError, match="image must be"):
            cc(img[:, :, :16], if_test=True)
    assert tuple(empty.shape) == (0, 7) and empty.dtype == full.dtype and empty.device == full.device
    assert tuple(feats.shape) == (0, arch.embed_dim) and tuple(dense.shape)[0::2] == (0, arch.embed_dim)
    assert torch.equal(one, full[2:3])


def test_train_caption_eval_entry_point(ops):
    from leclip_amd import train_caption
    out = train_caption.main(["--eval-only", "--trainer", "Caption_distill_double", "--backbone", "tiny", "--num-images", "48",
                              "MODEL.BACKBONE.PATH", "synthetic:1:cond", "INPUT.SIZE", "(32, 32)", "TRAINER.Caption.PREC", "fp32",
                              "DATALOADER.TEST.BATCH_SIZE", "16"])
    assert 0.0 < out["mAP"] <= 100.0


def test_module_surface_extras(ops, golden_dir):
    """Pieces of the reference surface not on the headline path: TextEncoder(if_sequence=True) (CDD.py:94-96),
    Transformer.forward in the reference's LND layout, the LayerNorm module, encode_text == TextEncoder on embeddings."""
    from leclip_amd.trainers import TextEncoder
    from oracle import clip_oracle as co
    sd = synth.make_state_dict(synth.TINY, seed=1, dist="cond")
    m = _build(synth.TINY, 1, "cond", torch.float32)
    t = np.load(os.path.join(golden_dir, "tokens_coco80.npz"))
    toks = torch.from_numpy(t["tokens_photo"][:4])
    te = TextEncoder(m)
    emb = sd["token_embedding.weight"][toks]
    seq = te(emb.to(DEV), toks.to(DEV), if_embedding=True, if_sequence=True)
    ref_seq = co.text_encoder(emb, toks, sd, if_sequence=True)
    np.testing.assert_allclose(seq.cpu().numpy(), ref_seq.numpy(), atol=2e-4, rtol=1e-4)
    pooled = te(emb.to(DEV), toks.to(DEV))
    np.testing.assert_allclose(pooled.cpu().numpy(), m.encode_text(toks.to(DEV)).cpu().numpy(), atol=1e-5, rtol=0)
    np.testing.assert_allclose(pooled.cpu().numpy(), co.encode_text(toks, sd).numpy(), atol=2e-4, rtol=1e-4)
    ids = te(toks.to(DEV), None, if_embedding=False)
    assert torch.equal(ids, pooled)
    # Transformer.forward: LND in, LND out (model.py:231-239), causal text stack
    x = (emb + sd["positional_embedding"]).permute(1, 0, 2).contiguous()
    y = m.transformer(x.to(DEV)).permute(1, 0, 2).cpu()
    xr = emb + sd["positional_embedding"]
    for i in range(synth.TINY.transformer_layers):
        xr = co.residual_block(xr, sd, f"transformer.resblocks.{i}.", 2, co.causal_mask(77))
    np.testing.assert_allclose(y.numpy(), xr.numpy(), atol=3e-4, rtol=3e-4)
    ln = m.ln_final(xr.to(DEV))
    np.testing.assert_allclose(ln.cpu().numpy(), co.layer_norm(xr, sd["ln_final.weight"], sd["ln_final.bias"]).numpy(), atol=2e-5, rtol=0)


def _rank_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from leclip_amd import parallel
    from leclip_amd.clip import build_model
    parallel.init_from_env(backend="gloo")       # 2 ranks share the one GPU of the test box: gloo carries the device tensors
    m = build_model(synth.make_state_dict(synth.TINY, seed=1)).float().to(DEV)
    toks = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), "golden", "tokens_coco80.npz"))["tokens_photo"][:8]).to(DEV)
    img = torch.from_numpy(synth.make_images(7, 32, seed=3)).to(DEV)        # ragged: 4 + 3
    sc = parallel.ShardedScorer(lambda x: m(x, toks)[0])
    got = sc.score_global(img)
    ref = m(img, toks)[0]
    q.put((rank, bool(torch.allclose(got, ref, atol=1e-5)), tuple(got.shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_scoring_two_ranks_on_device(ops):
    """parallel.ShardedScorer with the real HIP scorer on device tensors, world size 2 (both ranks on cuda:0, gloo
    transport): per-rank shards + all-gather of logits == unsharded logits."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == [(0, True, (7, 8)), (1, True, (7, 8))]


def test_score_postprocessing(ops, golden_dir):
    """Sliding-window aggregation and co-occurrence modulation against the outputs of the reference's own
    source lines (tests/golden/postprocess.npz: Caption_distill_double.py:614-618, 632-636, 654-660 executed on seeded scores
    and the reference's freq_stats.pkl), plus a second random case against the numpy restatement."""
    from oracle import metrics_oracle as mo
    g = np.load(os.path.join(golden_dir, "postprocess.npz"))
    got = ops.window_aggregate(torch.from_numpy(g["n2.output"]).to(DEV), torch.from_numpy(g["n2.output_blocks"]).to(DEV)).cpu().numpy()
    np.testing.assert_allclose(got, g["n2.output_final"], atol=1e-6, rtol=0)
    mn = ops.cooccurrence_matrix(g["freq.adj"], g["freq.nums"])
    np.testing.assert_allclose(mn.numpy(), g["n3.p"], atol=1e-7, rtol=0)
    got = ops.cooccurrence_adjust(torch.from_numpy(g["n3.output_pos_in"]).to(DEV), mn.to(DEV)).cpu().numpy()
    np.testing.assert_allclose(got, g["n3.output_pos_adjusted"], atol=2e-6, rtol=0)
    rng = np.random.RandomState(5)
    glob = rng.randn(37, 80).astype(np.float32) * 0.3
    blocks = rng.randn(37, 116, 80).astype(np.float32) * 0.3
    got = ops.window_aggregate(torch.from_numpy(glob).to(DEV), torch.from_numpy(blocks).to(DEV)).cpu().numpy()
    np.testing.assert_allclose(got, mo.window_aggregate(glob, blocks), atol=1e-6, rtol=0)


def test_crop_resize_is_bit_exact_with_pillow(ops, golden_dir):
    """leclip_crop_resize_fwd against Pillow's own bicubic resize + torchvision-style centre crop / ToTensor / Normalize
    (tests/golden/multicrop.npz, generated with Pillow in the build container): every float bit for bit - down- and
    up-scaling, reflect-padded rows, windows cut at the right edge; then every window of a 375x500 image against the
    numpy restatement of the resampler (oracle/multicrop_oracle.py), and the MultiCropper batch layout."""
    from leclip_amd import multicrop
    from oracle import multicrop_oracle as mc
    g = np.load(os.path.join(golden_dir, "multicrop.npz"))
    h, w = int(g["pil.src_hw"][0]), int(g["pil.src_hw"][1])
    src = synth.make_u8_image(h, w, seed=int(g["pil.src_seed"]))
    dsrc = torch.from_numpy(src).to(DEV)
    for i, win in enumerate(g["pil.windows"]):
        wt = torch.from_numpy(win[None, :5].astype(np.int32)).to(DEV)
        out = ops.crop_resize(dsrc, wt, int(win[5]), multicrop.CLIP_PIXEL_MEAN, multicrop.CLIP_PIXEL_STD)
        assert out.shape == (1, 1, 3, int(win[5]), int(win[5]))
        assert np.array_equal(out[0, 0].cpu().numpy(), g[f"pil.{i}.f32"]), i
    # all 570-ish windows of two images at once, S = 32 (cheap for the numpy oracle), fp32 exact and fp16 = rounded fp32
    src2 = np.stack([src, synth.make_u8_image(h, w, seed=6)])
    cropper = multicrop.MultiCropper(size=32)
    img, blocks = cropper(torch.from_numpy(src2).to(DEV))
    per_scale = multicrop.enumerate_windows(h, w)
    assert img.shape == (2, 3, 32, 32) and [b.shape[1] for b in blocks] == [len(p) for p in per_scale]
    for b in range(2):
        np.testing.assert_array_equal(img[b].cpu().numpy(), mc.transform_window(src2[b], multicrop.full_image_window(h, w)[0], 32, cropper.mean, cropper.std)[1])
        for blk, wins in zip(blocks, per_scale):
            got = blk[b].cpu().numpy()
            for j in range(0, len(wins), 7):      # every 7th window (the oracle is pure numpy)
                np.testing.assert_array_equal(got[j], mc.transform_window(src2[b], wins[j], 32, cropper.mean, cropper.std)[1], err_msg=f"{b} {wins[j]}")
    half = multicrop.MultiCropper(size=32, dtype=torch.float16)(torch.from_numpy(src2).to(DEV))[0]
    assert torch.equal(half, img.half())


@pytest.mark.parametrize("dt", DTYPES)
def test_image_tail_kernel(ops, dt):
    """leclip_image_tail_fwd (class-row gather + ln_post + projection + cosine logits in one launch, MFMA contractions)
    against the oracle's LayerNorm -> @ proj -> cosine_logits; ragged batch (B % 16 != 0), features-only and logits-only calls."""
    from oracle import clip_oracle as co
    for (b, t, d, e, c) in ((37, 5, 768, 512, 80), (16, 1, 1024, 768, 80), (3, 2, 128, 64, 7)):
        x = (_rand((b * t, d), 61, 2.0) + 0.3).to(dt)
        g, bt = _rand((d,), 62) * 0.1 + 1, _rand((d,), 63) * 0.1
        proj = _rand((d, e), 64, d ** -0.5).to(dt)
        txt = _rand((c, e), 65)
        hn = co.layer_norm(x.float().view(b, t, d)[:, 0], g, bt)
        if dt != torch.float32:
            hn = hn.to(dt).float()          # the MFMA's A operand is the LayerNorm output rounded to the compute dtype
        feat_ref = hn.double() @ proj.double()
        ref = co.cosine_logits(feat_ref, txt.double(), 4.0)
        feat, logits = ops.image_tail(x.to(DEV), b, t * d, g.to(DEV), bt.to(DEV), proj.t().contiguous().to(DEV), txt.to(DEV), 4.0, want_features=True)
        # 16-bit modes: the kernel's fp32 LayerNorm and the oracle's can round an element of h to neighbouring 16-bit values
        ftol = _tol(dt, 2e-5, 1e-3, 8e-3)
        np.testing.assert_allclose(feat.double().cpu().numpy(), feat_ref.numpy(), atol=ftol * float(feat_ref.abs().max()) + 1e-5, rtol=0)
        np.testing.assert_allclose(logits.double().cpu().numpy(), ref.numpy(), atol=_tol(dt, 5e-6, 2e-3, 1.6e-2), rtol=0)
        # the logit contraction itself is exact fp32 on the kernel's own features
        np.testing.assert_allclose(logits.double().cpu().numpy(), co.cosine_logits(feat.double().cpu(), txt.double(), 4.0).numpy(), atol=5e-6, rtol=0)
        f_only, none = ops.image_tail(x.to(DEV), b, t * d, g.to(DEV), bt.to(DEV), proj.t().contiguous().to(DEV))
        assert none is None and torch.equal(f_only, feat)
        none, l_only = ops.image_tail(x.to(DEV), b, t * d, g.to(DEV), bt.to(DEV), proj.t().contiguous().to(DEV), txt.to(DEV), 4.0)
        assert none is None and torch.equal(l_only, logits)


def test_logits_backward_and_row_moves(ops):
    """leclip_l2norm_logits_bwd against autograd through the oracle's cosine_logits; gather_rows / scatter_rows against indexing."""
    from oracle import clip_oracle as co
    for (b, c, d) in ((512, 80, 512), (7, 5, 64), (2500, 3, 768)):
        img, txt, dl = _rand((b, d), 71), _rand((c, d), 72), _rand((b, c), 73)
        tv = txt.clone().double().requires_grad_(True)
        (co.cosine_logits(img.double(), tv, 4.0) * dl.double()).sum().backward()
        got = ops.l2norm_logits_bwd(img.to(DEV), txt.to(DEV), dl.to(DEV), 4.0)
        np.testing.assert_allclose(got.double().cpu().numpy(), tv.grad.numpy(), atol=2e-5 * float(tv.grad.abs().max()), rtol=0)
    for dt in DTYPES:
        src = _rand((40, 64), 74).to(dt)
        idx = torch.tensor([3, 39, 0, 17], dtype=torch.int64)
        assert torch.equal(ops.gather_rows(src.to(DEV), idx.to(DEV)).cpu(), src[idx])
        sc = ops.scatter_rows(src[:4].contiguous().to(DEV), idx.to(DEV), 50).cpu()
        want = torch.zeros(50, 64, dtype=dt)
        want[idx] = src[:4]
        assert torch.equal(sc, want)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_fused_layernorm_on_large_mean_rows(ops, dt):
    """Rows whose mean dwarfs their spread (|mean| / std = 100) plus a few 1e2-magnitude outlier channels - the shape of real
    CLIP activations (massive channels): the epilogue's block partials are centred (sum, M2 about the block mean) and merged
    with the parallel-variance update, so the statistics lose nothing to E[x^2] - mean^2 cancellation; the folded form
    rstd * (x.W'^T - mean * colsum) is then checked against fp64 LayerNorm -> linear of the SAME 16-bit rows."""
    from leclip_amd.hip import engine
    g = torch.Generator(device="cpu").manual_seed(3)
    for (m, n, k) in ((700, 256, 768), (20000, 768, 768)):
        base = torch.randn(m, 1, generator=g) * 100.0                    # per-row offset, |mean| ~ 100
        x = base + torch.randn(m, k, generator=g)                        # std 1 around it
        x[:, [5, 77, 300]] += torch.tensor([150.0, -120.0, 90.0])        # outlier channels
        x = x.to(dt)
        xd = x.double()
        mu, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
        # (1) statistics from the producing GEMM's epilogue partials: identity weights make the GEMM output equal its residual input
        eye = torch.zeros(k, k, dtype=dt)
        part = torch.empty((k // 64, m, 2), dtype=torch.float32, device=DEV)
        y = ops.gemm_ln(torch.zeros(m, k, dtype=dt, device=DEV), eye.to(DEV), torch.zeros(k, device=DEV), residual=x.to(DEV), stats_out=part)
        assert torch.equal(y.cpu(), x)
        st = ops.ln_stats_finalize(part, k)
        np.testing.assert_allclose(st[:, 0].double().cpu().numpy(), mu[:, 0].numpy(), rtol=2e-6, atol=1e-4)
        np.testing.assert_allclose(st[:, 1].double().cpu().numpy(), (1 / torch.sqrt(var + 1e-5))[:, 0].numpy(), rtol=2e-4)
        st2 = ops.row_stats(x.to(DEV))
        np.testing.assert_allclose(st2[:, 1].double().cpu().numpy(), (1 / torch.sqrt(var + 1e-5))[:, 0].numpy(), rtol=2e-4)
        # (2) the folded LayerNorm GEMM on those rows
        gamma, beta = torch.randn(k, generator=g) * 0.1 + 1, torch.randn(k, generator=g) * 0.1
        w, b = torch.randn(n, k, generator=g) * k ** -0.5, torch.randn(n, generator=g) * 0.1
        wf, cs, cb = engine._fold_ln(w, b, gamma, beta, dt, DEV)
        ref = ((xd - mu) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()) @ w.double().t() + b.double()
        out = ops.gemm_ln(x.to(DEV), wf, cb, ln_stats=st, ln_colsum=cs, out_dtype=torch.float32)
        err = float((out.double().cpu() - ref).abs().max())
        # Yardstick: the UNFUSED path on the same rows - the LayerNorm kernel writes h = LN(x) rounded to 16 bits, a plain GEMM multiplies it
        # with W rounded to 16 bits.  The folded form rounds W' = gamma * W instead of h and W; both carry one 16-bit rounding per
        # product term, dominated by the 150-sigma channels (|x - mean| * rstd ~ 20 at eps 2^-11 / 2^-8).
        h = ops.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV))
        plain = ops.gemm(h, w.to(dt).to(DEV), b.to(DEV), out_dtype=torch.float32)
        err_plain = float((plain.double().cpu() - ref).abs().max())
        scale = float(ref.abs().max())
        print(f"fused-LN [{dt}, {m}x{n}x{k}]: folded err {err:.3e}, unfused (layernorm -> gemm) err {err_plain:.3e}, max|ref| {scale:.2f}")
        # bounds: 2 x the measured error of the folded path as a fraction of max|ref| (round 4, MI355X: fp16 2.1e-4 / 2.3e-4, bf16 2.1e-3 /
        # 1.6e-3 on the two shapes; the unfused path measured 4.5e-4 - 7.5e-4 and 3.4e-3 - 5.2e-3: the folded form is 2 - 3 x BETTER, it
        # rounds gamma * W once instead of h and W), and the folded form may never be worse than the unfused one
        assert err <= (4.5e-3 if dt == torch.bfloat16 else 5e-4) * scale, (err, scale)
        assert err <= err_plain, (err, err_plain)


def test_rccl_gather_path_single_rank(tmp_path):
    """The N>1 code path of parallel.py on the real backend: a one-rank RCCL ("nccl") process group in a child process,
    all_gather_into_tensor on device tensors through parallel._gather_flat, and ShardedScorer end to end."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, os.getcwd())
        from leclip_amd import parallel
        rank, world, local = parallel.init_from_env()
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        x = torch.arange(12, dtype=torch.float32, device="cuda").view(3, 4)
        y = parallel._gather_flat(x, 1)
        assert torch.equal(x, y)
        sc = parallel.ShardedScorer(lambda im: im.float().mean(dim=(1, 2, 3))[:, None].expand(-1, 80).contiguous())
        out = sc.score_global(torch.ones(5, 3, 8, 8, device="cuda"))
        assert out.shape == (5, 80) and float(out.mean()) == 1.0
        dist.barrier(); dist.destroy_process_group()
        print("rccl ok")
    """)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stderr[-2000:]


def test_full_batch_forward_is_bit_reproducible(ops):
    """Race screen for the persistent GEMM (K-loop pipelined across tiles, counted vmcnt waits relaxed past the epilogue's
    stores) and the pipelined attention at the benchmark shape: B=256 bf16 image features, repeated, must be identical."""
    m = _build(synth.VIT_B16, 0, "cond", torch.bfloat16)
    img = torch.from_numpy(synth.make_images(256, 224, seed=77)).to(DEV)
    with torch.no_grad():
        ref = m.encode_image(img).clone()
        assert bool(torch.isfinite(ref).all())
        for _ in range(25):
            assert torch.equal(m.encode_image(img), ref)


def test_bench_self_launches_two_ranks_on_one_gpu():
    """The driver's invocation for N > 1 - a bare `python bench.py --gpus 2 ...` - end to end on this one-GPU box: the script starts its
    own two ranks under torch.distributed.run (child process, 127.0.0.1), both on cuda:0 with the gloo transport (RCCL refuses two ranks
    on one device; LECLIP_DIST_BACKEND is recorded in the line), every rank scores its 32 images, the per-rank logits are all-gathered,
    rank 0 prints ONE JSON line for the whole job."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["LECLIP_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "32",
                          "--no-cpu-baseline", "--no-second-dtype", "--profile-every", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "img/s"
    assert d["config"]["global_batch"] == 64 and "allgather" in d["config"]["parallelism"] and d["value"] > 0
    assert "LECLIP_DIST_BACKEND=gloo" in d["env_overrides"]
    assert abs(d["value"] - 64 * 3 / (d["ms_per_step"] * 1e-3 * 3)) / d["value"] < 1e-6
    assert d["config"]["world_size"] == 2 and d["config"]["backend"] == "gloo" and "mAP_gathered" not in d     # (B = 32: not the fixture's workload)
    # BASELINE configs[3]'s "eval mAP" on the all-gathered logits, at the benchmark's own batch: two ranks x 256 images, scored by rank 0 against
    # the committed reference logits of exactly those 512 images (tests/golden/vitb16_cfg4_logits.npz) - no CPU oracle in the loop
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-second-dtype", "--profile-every", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    m = d["mAP_gathered"]
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 512 and m["n_images"] == 512
    assert abs(m["delta"]) <= 0.2 and m["accuracy_gate"].startswith("met") and m["max_abs_logit_diff"] <= 4e-3 and m["top1_agree"] >= 0.98


def test_caption_feature_mixing_kernel(ops, golden_dir):
    """leclip_topk_mix_fwd (exact-fp32 similarity GEMM + native top-10 / mean / average) against the reference's own lines :444-448
    (tests/golden/caption_branch.npz mix.*), incl. an exact tie in the candidate set and a table that is not a multiple of 64 rows."""
    g = np.load(os.path.join(golden_dir, "caption_branch.npz"))
    img, cap = torch.from_numpy(g["mix.image_feature"]).to(DEV), torch.from_numpy(g["mix.caption_text_feats"]).to(DEV)
    got = ops.topk_mix(img, cap, 10)
    np.testing.assert_allclose(got.cpu().numpy(), g["mix.mixed"], atol=2e-6, rtol=0)
    ref1 = torch.from_numpy(g["mix.image_feature"])
    sim = ref1 @ torch.from_numpy(g["mix.caption_text_feats"]).t()
    want = 0.5 * (ref1 + torch.from_numpy(g["mix.caption_text_feats"])[sim.topk(1, -1).indices[:, 0]])
    np.testing.assert_allclose(ops.topk_mix(img, cap, 1).cpu().numpy(), want.numpy(), atol=2e-6, rtol=0)


def test_dense_clip_with_caption_features(ops, golden_dir):
    """DenseCLIP test branch with the reference's caption-feature mixing switched on (TEST.caption_text_feats / set_caption_text_feats):
    the table comes from the model's own caption_features() (generate_caption_text_features.py:82-88), scores against the oracle."""
    from leclip_amd.clip import build_model
    from leclip_amd.config import get_cfg_default
    from leclip_amd.datasets import coco_object_categories
    from leclip_amd.trainers import DenseCLIP
    from oracle import clip_oracle as co
    arch = synth.TINY
    sd = synth.make_state_dict(arch, seed=1, dist="cond")
    cfg = get_cfg_default()
    cfg.INPUT.SIZE = (arch.image_resolution, arch.image_resolution)
    model = DenseCLIP(cfg, coco_object_categories, build_model(sd).float()).to(DEV).eval()
    t = np.load(os.path.join(golden_dir, "tokens_coco80.npz"))
    caps = torch.from_numpy(t["tokens_photo"])
    table = model.caption_features(caps)
    ref_table = co.caption_text_features(caps, sd)
    np.testing.assert_allclose(table.cpu().numpy(), ref_table.numpy(), atol=2e-5, rtol=0)
    model.set_caption_text_feats(ref_table)
    img = torch.from_numpy(synth.make_images(5, arch.image_resolution, seed=3))
    with torch.no_grad():
        logits_, logits_local, _, _, _ = model(img.to(DEV), if_test=True)
    toks = torch.from_numpy(t["tokens_ctx16"])
    prefix, suffix = co.prompt_buffers(toks, sd, 16)
    pl = model.prompt_learner
    ref, ref_local = co.dense_clip_forward(img, sd, pl.ctx.detach().cpu(), pl.ctx_double.detach().cpu(), None, prefix, suffix, toks, 40.0, 4.0,
                                           caption_text_feats=ref_table)
    np.testing.assert_allclose(logits_.cpu().numpy(), ref.numpy(), atol=2e-4, rtol=0)
    np.testing.assert_allclose(logits_local.cpu().numpy(), ref_local.numpy(), atol=2e-4, rtol=1e-4)
    model.set_caption_text_feats(None)
    with torch.no_grad():
        plain = model(img.to(DEV), if_test=True)[0]
    assert float((plain - logits_).abs().max()) > 1e-3          # the mixing does change the global scores


def _eval_batches_device():
    g = torch.Generator().manual_seed(17)
    out = []
    for b, scales in ((5, (3, 4)), (1, (3, 4)), (4, (3, 4))):
        lab = (torch.rand(b, 80, generator=g) < 0.1).long()
        img = torch.from_numpy(synth.make_images(b, 32, seed=100 + b))
        blocks = [torch.from_numpy(synth.make_images(b * w, 32, seed=200 + 10 * b + w)).reshape(b, w, 3, 32, 32) for w in scales]
        out.append({"img": img, "label": lab, "img_blocks": blocks})
    return out


def _eval_trainer_device():
    from leclip_amd.config import get_cfg_default
    from leclip_amd.registry import build_evaluator, build_trainer
    cfg = get_cfg_default()
    cfg.merge_from_list(["MODEL.BACKBONE.NAME", "tiny", "MODEL.BACKBONE.PATH", "synthetic:1:cond", "INPUT.SIZE", "(32, 32)", "TRAINER.Caption.PREC", "fp32",
                         "TRAIN.MODEL", "DenseCLIP", "DATALOADER.TEST.BATCH_SIZE", "4"])
    torch.manual_seed(3)
    tr = build_trainer(cfg, evaluator=build_evaluator(cfg))
    tr.test_loader = _eval_batches_device()
    return tr


def _eval_rank_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from leclip_amd import parallel
    parallel.init_from_env(backend="gloo")
    tr = _eval_trainer_device()
    value = tr.test()
    q.put((rank, float(value), tr.evaluator.evaluate()))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_test_sharded_two_ranks_on_device(ops):
    """Caption_distill_double.test() sharded over two ranks (both on cuda:0, gloo transport) with the real DenseCLIP scorer: global +
    local scores, sliding windows of two scales split by rank, a one-image batch - every metric equals the single-process run exactly
    (window scores are batch-invariant bit for bit; max / min over windows combine exactly)."""
    import socket
    import torch.multiprocessing as mp
    single = _eval_trainer_device()
    want = float(single.test())
    want_all = single.evaluator.evaluate()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_eval_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, value, all_metrics in res:
        assert value == want and all_metrics == want_all, (rank, value, want)
