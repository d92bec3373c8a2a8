"""Tower engines: weight packing and the kernel sequence of the two CLIP towers.

Data layout in HBM (all row-major, token-major within an image / prompt):

* residual stream  ``x``   [B*T, d]      compute dtype (bf16 / fp16 / fp32), updated in place by the
                                         out-proj and c_proj GEMM epilogues (bias + residual fused);
* LayerNorm output ``h``   [B*T, d]      compute dtype, the A operand of the next GEMM;
* packed QKV       ``qkv`` [B*T, 3d]     compute dtype, columns q|k|v in in_proj order; attention reads the
                                         128-byte (b, t, head) rows in place - no head-major re-layout;
* attention output ``ctx`` [B*T, d]      compute dtype; MLP hidden ``u`` [B*T, 4d] (QuickGELU fused in c_fc);
* weights [N, K] (nn.Linear layout, K contiguous) in the compute dtype; biases, LayerNorm affine,
  class / positional embeddings, token table in fp32.

One forward of a tower with L blocks is 1 + 7L + 1 kernel launches (patch-embed adds 2).  The text tower and small image
batches enqueue them on the caller's current HIP stream.  A large image batch (``VisionEngine.streams`` parts, B >= 128) runs
as contiguous parts, EVERY part on an engine-owned side stream shared per device (never the caller's): the side streams wait for
the caller's stream before the first launch (fork) and the caller's stream waits for them after the last (join), so to the
caller the forward is ordered like any other work on its stream.  Nothing synchronises the host and all workspaces are cached
per (batch size, stream).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from ._capi import ACT_NONE, ACT_QUICKGELU



def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class _Block:
    __slots__ = ("ln1_w", "ln1_b", "w_qkv", "b_qkv", "w_o", "b_o", "ln2_w", "ln2_b", "w_fc", "b_fc", "w_pr", "b_pr",
                 "wf_qkv", "cs_qkv", "cb_qkv", "wf_fc", "cs_fc", "cb_fc")


def _fold_ln(w: torch.Tensor, bias: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, dtype: torch.dtype, device=None):
    """LayerNorm folded into the consuming linear layer: W' = W * gamma (rounded once to the compute dtype),
    colsum[n] = sum_k W'[n,k] (of the ROUNDED weights the MFMA will see), c[n] = sum_k beta[k] W[n,k] + b[n].
    One-off host-side packing arithmetic (fp64 on the CPU); nothing here runs per forward."""
    device = device if device is not None else w.device
    wq = w.detach().cpu().double()
    g, bt, b0 = gamma.detach().cpu().double(), beta.detach().cpu().double(), bias.detach().cpu().double()
    wf = (wq * g[None, :]).float().to(dtype)
    colsum = wf.double().sum(dim=1).float()
    cb = (wq.float().to(dtype).double() @ bt + b0).float()
    return wf.contiguous().to(device), colsum.contiguous().to(device), cb.contiguous().to(device)


def pack_blocks(resblocks, dtype: torch.dtype, device):
    """Pack the parameters of a ``Transformer``'s blocks (clip/model.py:207-228 layout)."""
    out = []
    for blk in resblocks:
        p = _Block()
        p.ln1_w, p.ln1_b = _f32(blk.ln_1.weight, device), _f32(blk.ln_1.bias, device)
        p.ln2_w, p.ln2_b = _f32(blk.ln_2.weight, device), _f32(blk.ln_2.bias, device)
        p.w_qkv = blk.attn.in_proj_weight.detach().to(device=device, dtype=dtype).contiguous()
        p.b_qkv = _f32(blk.attn.in_proj_bias, device)
        p.w_o = blk.attn.out_proj.weight.detach().to(device=device, dtype=dtype).contiguous()
        p.b_o = _f32(blk.attn.out_proj.bias, device)
        p.w_fc = blk.mlp.c_fc.weight.detach().to(device=device, dtype=dtype).contiguous()
        p.b_fc = _f32(blk.mlp.c_fc.bias, device)
        p.w_pr = blk.mlp.c_proj.weight.detach().to(device=device, dtype=dtype).contiguous()
        p.b_pr = _f32(blk.mlp.c_proj.bias, device)
        if dtype != torch.float32:
            p.wf_qkv, p.cs_qkv, p.cb_qkv = _fold_ln(blk.attn.in_proj_weight, p.b_qkv, p.ln1_w, p.ln1_b, dtype, device)
            p.wf_fc, p.cs_fc, p.cb_fc = _fold_ln(blk.mlp.c_fc.weight, p.b_fc, p.ln2_w, p.ln2_b, dtype, device)
        out.append(p)
    return out


class _Workspace:
    """Activation buffers of one tower for one (batch, tokens) shape."""

    def __init__(self, rows: int, d: int, dtype: torch.dtype, device, batch: int = 0):
        self.h = torch.empty((rows, d), dtype=dtype, device=device)
        self.qkv = torch.empty((rows, 3 * d), dtype=dtype, device=device)
        self.ctx = torch.empty((rows, d), dtype=dtype, device=device)
        self.u = torch.empty((rows, 4 * d), dtype=dtype, device=device)
        # fused-LayerNorm path: per-row (mean, rstd) and the epilogue's partial sums per 64-column block
        self.stats = torch.empty((rows, 2), dtype=torch.float32, device=device)
        # slot-major [d/64][rows][2]: the rows a wave finishes together are contiguous bytes of one slot (one full-line store per pass)
        self.partials = torch.empty((d // 64, rows, 2), dtype=torch.float32, device=device)
        # in-producer LayerNorm merge (ops.gemm_res_stats): one arrival counter per 384-row block, zero between launches
        self.tickets = torch.zeros(((rows + 383) // 384 + 1,), dtype=torch.int32, device=device)
        if batch > 0:   # image tower: the last block's class-token rows (run_blocks, ``cls_last``)
            self.cls_x = torch.empty((batch, d), dtype=dtype, device=device)
            self.cls_u = torch.empty((batch, 4 * d), dtype=dtype, device=device)
            self.cls_stats = torch.empty((batch, 2), dtype=torch.float32, device=device)
            self.cls_partials = torch.empty((d // 64, batch, 2), dtype=torch.float32, device=device)
            self.cls_index = (torch.arange(batch, device=device, dtype=torch.int64) * (rows // batch)).contiguous()
            # the class rows' pairs inside the flat [d/64 * rows, 2] view of the partials: slot * rows + class row
            self.cls_pair_index = (torch.arange(d // 64, device=device, dtype=torch.int64)[:, None] * rows + self.cls_index[None, :]).reshape(-1).contiguous()


class _Walk:
    """Walk-order policy through a residual block's launches (leclip_set_walk_order: which rows a kernel takes first, never what it
    computes).  A consumer that walks its rows in the order opposite to its producer's starts on the rows written last, which the 256 MiB
    memory-side cache still holds.  Measured on MI355X at B = 256 (profiles/r04_walk_order.txt): c_proj, which reads the 310 MB MLP hidden,
    gains 4 % from it with the batch as ONE part (259 -> 249 us), but qkv loses as much when its producer walks descending, and with the
    batch as two stream parts - the product schedule - no policy moves the step time (-0.6 .. +0.1 %).  Policies (A/B: bench.py --walk):
    "default" (library order for every launch; the engine's choice), "c_proj", "c_fc" (that kernel descending), "alternate"."""
    ORDER = {"default": {}, "c_proj": {"c_proj": 1}, "c_fc": {"c_fc": 1},
             "alternate": None}

    def __init__(self, policy: str):
        self.table = self.ORDER[policy]
        self.on = policy != "default"
        self.flip = 1

    def set(self, kernel: str):
        if not self.on:
            return
        if self.table is None:
            ops.set_walk_order(self.flip)
            self.flip ^= 1
        else:
            ops.set_walk_order(self.table.get(kernel, -1))

    def done(self):
        if self.on:
            ops.set_walk_order(-1)


def run_blocks(x: torch.Tensor, blocks, ws: _Workspace, batch: int, tokens: int, heads: int, causal: bool,
               taps: Optional[dict] = None, fuse_ln: Optional[bool] = None, have_partials: bool = False,
               cls_last: bool = False, walk: str = "default", producer_merge: bool = True) -> torch.Tensor:
    """x [B*T, d] (updated in place) through the residual attention blocks (clip/model.py:225-228).

    16-bit modes fuse both LayerNorms of a block into the GEMM that consumes them (``fuse_ln``): the producing GEMM's
    epilogue emits per-row block partials (sum, M2 about the block mean per 64 columns; slot-major [d/64][rows][2]), a 5 us merge kernel (launched by
    the consumer's C entry point) turns them into (mean, rstd), and the consuming GEMM - fed the raw residual rows and
    gamma-folded weights - normalises in its epilogue: LN(x) is never written to HBM nor rounded to 16 bits.
    ``have_partials``: ``ws.partials`` already holds the partials of ``x`` (written by the fused
    patch-embedding / ln_pre pass); otherwise one read-only pass over x provides the first statistics.  The fp32 parity mode and
    the per-stage taps keep LayerNorm as its own kernel.

    ``cls_last`` (image tower, fused path): VisionTransformer.forward keeps ``x[:, 0, :]`` alone (clip/model.py:271), so in the LAST
    block only the class token's row has a consumer.  Keys and values still need every token (the k|v two thirds of the qkv GEMM run on
    all rows, the q third on the class rows), the attention computes the first query block only, and out-proj, LayerNorm 2, c_fc,
    QuickGELU and c_proj run on the B class rows (read in place at their stride T*d, written compactly).  Rows are independent in every one of these kernels and both GEMM
    families produce the same bits, so the class rows equal the full computation's bit for bit
    (tests/test_gpu_parity.py::test_full_batch_properties); returned is the compact [B, d] class-row block instead of x."""
    d = x.shape[1]
    if fuse_ln is None:
        fuse_ln = x.dtype != torch.float32 and taps is None and d % 64 == 0
    if fuse_ln:
        if have_partials:
            src = dict(ln_partials=ws.partials, ln_stats_ws=ws.stats)
        else:
            ops.row_stats(x, out=ws.stats)
            src = dict(ln_stats=ws.stats)
        # Round 5: out-proj / c_proj FINISH the statistics of the rows they write (ops.gemm_res_stats: in the launch itself on the 384 x 256
        # kernel, by the merge kernel behind the GEMM otherwise) - the consumer takes (mean, rstd) as they are, no merge launch in front of it
        after = dict(ln_stats=ws.stats)
        tickets = ws.tickets if producer_merge else None     # (None: always the merge launch - A/B, bench.py --no-producer-merge)
        last = len(blocks) - 1
        walk = _Walk(walk)
        try:   # (an exception mid-block must not leave the calling thread's walk-order hint set for later launches)
            for i, p in enumerate(blocks):
                if cls_last and i == last:
                    walk.done()
                    x_c = x.view(batch, tokens * d)[:, :d]             # class rows in place: [B, d] with row stride T*d
                    cls = dict(ln_partials=ws.cls_partials, ln_stats_ws=ws.cls_stats)
                    if i > 0:
                        # keys and values of every token (the k|v rows of in_proj: N = 2d), queries of the class rows only (their statistics:
                        # the class rows of the (mean, rstd) the previous block's c_proj finished)
                        ops.gemm_ln(x, p.wf_qkv[d:], p.cb_qkv[d:], ln_colsum=p.cs_qkv[d:], out=ws.qkv[:, d:], **src)
                        ops.gather_rows(ws.stats, ws.cls_index, out=ws.cls_stats)
                        ops.gemm_ln(x_c, p.wf_qkv[:d], p.cb_qkv[:d], ln_colsum=p.cs_qkv[:d], out=ws.qkv.view(batch, tokens * 3 * d)[:, :d], ln_stats=ws.cls_stats)
                    elif "ln_partials" in src:
                        ops.gemm_ln(x, p.wf_qkv[d:], p.cb_qkv[d:], ln_colsum=p.cs_qkv[d:], out=ws.qkv[:, d:], **src)
                        ops.gather_rows(ws.partials.view(-1, 2), ws.cls_pair_index, out=ws.cls_partials.view(-1, 2))
                        ops.gemm_ln(x_c, p.wf_qkv[:d], p.cb_qkv[:d], ln_colsum=p.cs_qkv[:d], out=ws.qkv.view(batch, tokens * 3 * d)[:, :d], **cls)
                    else:   # (a one-block tower without fused patch statistics: the whole qkv GEMM)
                        ops.gemm_ln(x, p.wf_qkv, p.cb_qkv, ln_colsum=p.cs_qkv, out=ws.qkv, **src)
                    ops.attention(ws.qkv, batch, tokens, heads, causal, out=ws.ctx, q_rows=1)
                    ctx_c = ws.ctx.view(batch, tokens * d)[:, :d]
                    ops.gemm_ln(ctx_c, p.w_o, p.b_o, residual=x_c, stats_out=ws.cls_partials, out=ws.cls_x)
                    ops.gemm_ln(ws.cls_x, p.wf_fc, p.cb_fc, ln_colsum=p.cs_fc, act=ACT_QUICKGELU, out=ws.cls_u, **cls)
                    ops.gemm(ws.cls_u, p.w_pr, p.b_pr, residual=ws.cls_x, out=ws.cls_x)
                    return ws.cls_x
                walk.set("qkv")
                ops.gemm_ln(x, p.wf_qkv, p.cb_qkv, ln_colsum=p.cs_qkv, out=ws.qkv, **src)
                walk.set("attention")
                ops.attention(ws.qkv, batch, tokens, heads, causal, out=ws.ctx)
                walk.set("out_proj")
                ops.gemm_res_stats(ws.ctx, p.w_o, p.b_o, residual=x, partials=ws.partials, stats_out=ws.stats, tickets=tickets, out=x)
                walk.set("c_fc")
                ops.gemm_ln(x, p.wf_fc, p.cb_fc, ln_colsum=p.cs_fc, act=ACT_QUICKGELU, out=ws.u, **after)
                walk.set("c_proj")
                if i < last:
                    ops.gemm_res_stats(ws.u, p.w_pr, p.b_pr, residual=x, partials=ws.partials, stats_out=ws.stats, tickets=tickets, out=x)
                    src = after
                else:
                    ops.gemm(ws.u, p.w_pr, p.b_pr, residual=x, out=x)
        finally:
            walk.done()
        return x
    for i, p in enumerate(blocks):
        ops.layernorm(x, p.ln1_w, p.ln1_b, out=ws.h)
        ops.gemm(ws.h, p.w_qkv, p.b_qkv, out=ws.qkv)
        ops.attention(ws.qkv, batch, tokens, heads, causal, out=ws.ctx)
        if taps is not None:
            taps[f"block{i}.ln_1"] = ws.h.float().clone()
            taps[f"block{i}.attn_ctx"] = ws.ctx.float().clone()
        ops.gemm(ws.ctx, p.w_o, p.b_o, residual=x, out=x)
        ops.layernorm(x, p.ln2_w, p.ln2_b, out=ws.h)
        ops.gemm(ws.h, p.w_fc, p.b_fc, act=ACT_QUICKGELU, out=ws.u)
        if taps is not None:
            taps[f"block{i}.after_attn"] = x.float().clone()
            taps[f"block{i}.ln_2"] = ws.h.float().clone()
            taps[f"block{i}.gelu"] = ws.u.float().clone()
        ops.gemm(ws.u, p.w_pr, p.b_pr, residual=x, out=x)
        if taps is not None:
            taps[f"block{i}.out"] = x.float().clone()
    return x


def round_fill(batch: int, tokens: int, width: int, n_cu: int) -> float:
    """Fraction of the CU-rounds of the four block GEMMs that carry a tile when the batch runs as one part (256 x 256 tiles, one
    persistent workgroup per CU; weights: K)."""
    if width % 256:
        return 1.0
    tile_rows = (batch * tokens + 255) // 256
    used = total = 0.0
    for n, k in ((3 * width, 1), (width, 1), (4 * width, 1), (width, 4)):
        tiles = tile_rows * (n // 256)
        used += k * tiles / n_cu
        total += k * ((tiles + n_cu - 1) // n_cu)
    return used / total


def stream_parts(batch: int, streams: int, split_sizes, split_min_batch: int, tokens: int, width: int, n_cu: int):
    """[(lo, hi), ...] row ranges of the parts a batch runs as on HIP streams of their own, or None for one piece.  Explicit
    ``split_sizes`` win; otherwise an even split into ``streams`` parts from ``split_min_batch`` images up, and for 64 <= batch <
    split_min_batch only where the tile rounds of the block GEMMs are filled below 0.8 (measured: B=64 gains 16 %, B=96 loses 6 %)."""
    if split_sizes is not None:
        sizes = [int(v) for v in split_sizes]
        if sum(sizes) != batch:
            raise ValueError(f"split_sizes {sizes} do not add up to the batch {batch}")
    elif streams > 1 and (batch >= split_min_batch or (batch >= 64 and round_fill(batch, tokens, width, n_cu) < 0.8)):
        n = min(streams, batch)
        sizes = [batch // n + (1 if i < batch % n else 0) for i in range(n)]
    else:
        return None
    sizes = [v for v in sizes if v > 0]
    if len(sizes) < 2:
        return None
    bounds, lo = [], 0
    for v in sizes:
        bounds.append((lo, lo + v))
        lo += v
    return bounds


_PART_STREAMS: Dict[int, list] = {}


def _part_streams(device, n: int):
    """The HIP streams the parts of a batch run on: ONE set per device for the whole process, none of them the caller's.  The runtime
    has a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and hands them to streams as they come into use; two parts whose
    streams share a queue run one after the other (measured: a second engine with streams of its own fell from 26.5 k to 21.2 k img/s,
    the rate of GPU_MAX_HW_QUEUES=1), so every engine uses the same few streams."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    pool = _PART_STREAMS.setdefault(idx, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


_HWQ_NOTED = False


def _note_hw_queues():
    """Say once when the stream parts run without the hardware-queue setting in effect (leclip_amd.configure() not called before the
    HIP runtime started and GPU_MAX_HW_QUEUES not set by the user): results are identical, but the parts may share a hardware queue and
    then run one after the other (measured 22.2 k instead of 26 k img/s at GPU_MAX_HW_QUEUES=1)."""
    global _HWQ_NOTED
    if _HWQ_NOTED:
        return
    _HWQ_NOTED = True
    import leclip_amd
    state = leclip_amd.hw_queue_state()
    if state in ("late", "default"):
        import warnings
        warnings.warn("leclip_amd: stream parts are running with the HIP runtime's default hardware queues (" + state + "); call "
                      "leclip_amd.configure() before the first device call, or set GPU_MAX_HW_QUEUES=8, for the overlapped schedule",
                      RuntimeWarning, stacklevel=3)


class VisionEngine:
    """VisionTransformer.forward (clip/model.py:259-276) as a HIP kernel sequence."""

    def __init__(self, visual, dtype: torch.dtype, device):
        self.dtype, self.device = dtype, device
        w = visual.conv1.weight.detach()
        self.width, _, self.patch, _ = w.shape
        self.resolution = int(visual.input_resolution)
        self.heads = self.width // 64
        grid = self.resolution // self.patch
        self.tokens = grid * grid + 1
        k = 3 * self.patch * self.patch
        align = 32 if dtype == torch.float32 else 64
        kp = (k + align - 1) // align * align
        wp = torch.zeros((self.width, kp), dtype=dtype, device=device)
        wp[:, :k] = w.reshape(self.width, k).to(device=device, dtype=dtype)
        self.wp = wp
        self.cls = _f32(visual.class_embedding, device)
        self.pos = _f32(visual.positional_embedding, device)
        self.ln_pre_w, self.ln_pre_b = _f32(visual.ln_pre.weight, device), _f32(visual.ln_pre.bias, device)
        self.ln_post_w, self.ln_post_b = _f32(visual.ln_post.weight, device), _f32(visual.ln_post.bias, device)
        self.proj = visual.proj.detach().to(device=device, dtype=dtype).contiguous()
        self.proj_t = self.proj.t().contiguous()   # [E, d]: K-contiguous B operand of the tail kernel's projection
        self.blocks = pack_blocks(visual.transformer.resblocks, dtype, device)
        self._ws: Dict[int, tuple] = {}
        # Large batches run as `streams` contiguous parts, every part on a per-device shared side stream forked from and joined to the
        # caller's stream (_on_streams; none of the parts runs on the caller's stream itself): every big GEMM
        # is a persistent grid of one workgroup per CU whose last round of tiles leaves most CUs idle (ViT-B/16, B=256: out-proj
        # and c_proj are 2.31 rounds), and the other part's kernels fill those CUs.  Images are independent and every kernel
        # is batch-invariant bit for bit, so the split does not change a single output value (tests/test_gpu_parity.py).
        # Measured (ViT-B/16, img/s one part -> two): B=256 24.1k -> 26.0k, 192 24.6k -> 26.1k, 128 20.4k -> 22.3k, 64 17.6k -> 20.5k,
        # but 96 24.3k -> 22.8k and 32 15.1k -> 12.1k; ViT-L/14@336 B=128 2 357 -> 2 441.
        self.streams = 2
        # forward() / score() consume the class token alone: the last block computes only what that row needs (run_blocks, cls_last:
        # the whole qkv GEMM, attention for the first query block, out-proj / MLP on B rows instead of B*T) - same bits, 6 % less time.
        self.cls_last_block = True
        self.split_min_batch = 128        # from here on two parts always paid; below, only when the tile rounds are badly filled
        self.split_sizes = None           # experiments: explicit part sizes instead of an even split
        # walk-order policy of the block kernels (run_blocks, _Walk): which rows a kernel takes first - same bits either way.  "default":
        # with the batch as two stream parts (the product schedule) no policy measured a gain (profiles/r04_walk_order.txt)
        self.walk = "default"
        # out-proj / c_proj merge their LayerNorm partials inside the launch (384 x 256 kernel; False: a merge launch behind every one of them)
        self.producer_merge = True
        self.beside_result = None         # what the last forward(..., beside=fn) call's fn returned

    def _parts(self, image: torch.Tensor):
        """Row ranges of the stream parts, or None when the batch runs as one piece."""
        if not image.is_cuda:
            return None
        n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
        return stream_parts(image.shape[0], self.streams, self.split_sizes, self.split_min_batch, self.tokens, self.width, n_cu)

    def _on_streams(self, image: torch.Tensor, fn, beside=None):
        """fn(image part) -> tuple of per-image tensors (or None entries); parts run concurrently, results are concatenated.
        ``beside``: a callable enqueued on the CALLER's stream after the parts were forked and before they are joined - work that does
        not depend on the image tower (the prompt-tuning step's text tower: trainers/caption_distill_double.py CustomCLIP.forward) runs
        beside it instead of behind it; its result is stored in ``self.beside_result``."""
        bounds = self._parts(image)
        if bounds is None:
            res = fn(image)
            if beside is not None:
                self.beside_result = beside()
            return res
        cur = torch.cuda.current_stream(self.device)
        side = _part_streams(self.device, len(bounds))
        _note_hw_queues()
        for st in side:
            st.wait_stream(cur)                      # fork BEFORE any part is enqueued: the inputs are ready on the caller's stream
        outs = []
        for st, (lo, hi) in zip(side, bounds):
            with torch.cuda.stream(st):
                res = fn(image[lo:hi])
            for t in res:
                if t is not None:
                    t.record_stream(cur)             # allocated on the part's stream, consumed on the caller's
            outs.append(res)
        if beside is not None:
            self.beside_result = beside()            # on the caller's stream, which the parts do not use
        for st in side:
            cur.wait_stream(st)
        return tuple(None if outs[0][k] is None else torch.cat([o[k] for o in outs], dim=0) for k in range(len(outs[0])))

    def _workspace(self, batch: int):
        # one workspace per (batch size, HIP stream): forwards issued on different streams (batch halves overlapping each
        # other's kernel tails) must not share activation buffers
        key = (batch, torch.cuda.current_stream(self.device).cuda_stream)
        if key not in self._ws:
            if len(self._ws) >= 8:
                self._ws.clear()
            rows = batch * self.tokens
            self._ws[key] = (
                _Workspace(rows, self.width, self.dtype, self.device, batch=batch),
                torch.empty((rows, self.width), dtype=self.dtype, device=self.device),
                torch.empty(ops._capi.load().leclip_patch_embed_ln_workspace_bytes(batch, self.resolution, self.patch, self.width,
                                                                                   ops.dtype_code(self.dtype)), dtype=torch.uint8, device=self.device),
                (torch.arange(batch, device=self.device, dtype=torch.int64) * self.tokens).contiguous(),
            )
        return self._ws[key]

    def _check_image(self, image: torch.Tensor):
        if image.dim() != 4 or image.shape[1] != 3 or image.shape[2] != self.resolution or image.shape[3] != self.resolution:
            raise ValueError(f"image must be [B,3,{self.resolution},{self.resolution}], got {tuple(image.shape)}")
        if not image.is_cuda:
            raise ValueError("image must live on the HIP device (there is no CPU execution path)")

    def _trunk(self, image: torch.Tensor, taps: Optional[dict] = None, cls_only: bool = False):
        """-> (rows, batch, class-row indices, class-row stride).  ``cls_only``: the caller consumes the class token alone, so the
        last block may compute just that (``self.cls_last_block``); rows is then the compact [B, d] class-row block."""
        self._check_image(image)
        batch = image.shape[0]
        ws, x, patch_ws, cls_rows = self._workspace(batch)
        image = image.contiguous()
        have_partials = False
        if taps is None and self.width % 64 == 0 and self.width <= 4096:
            # patch GEMM with the plain 16-bit epilogue, then class token + positional add + ln_pre in one row-wise pass that also
            # emits the block partials the first block's fused LayerNorm merges (16-bit modes, width <= 1024)
            have_partials = self.dtype != torch.float32 and self.width <= 1024
            ops.patch_embed_ln(image, self.wp, self.cls, self.pos, self.ln_pre_w, self.ln_pre_b, self.patch, self.dtype, workspace=patch_ws,
                               out=x.view(batch, self.tokens, self.width), stats_out=ws.partials if have_partials else None)
        else:
            ops.patch_embed(image, self.wp, self.cls, self.pos, self.patch, self.dtype, workspace=patch_ws,
                            out=x.view(batch, self.tokens, self.width))
            if taps is not None:
                taps["embed"] = x.float().clone()
            ops.layernorm(x, self.ln_pre_w, self.ln_pre_b, out=x)
            if taps is not None:
                taps["ln_pre"] = x.float().clone()
        cls_last = (cls_only and self.cls_last_block and taps is None and self.dtype != torch.float32 and self.width % 64 == 0
                    and self.width <= 1024 and self.proj.shape[1] % 16 == 0)     # (the conditions of the fused path and of the tail kernel)
        rows = run_blocks(x, self.blocks, ws, batch, self.tokens, self.heads, False, taps, have_partials=have_partials, cls_last=cls_last,
                          walk=self.walk, producer_merge=self.producer_merge)
        if cls_last:
            return rows, batch, None, self.width
        return x, batch, cls_rows, self.tokens * self.width

    def _tail(self, x: torch.Tensor, batch: int, cls_rows, row_stride: int, text_features: Optional[torch.Tensor], scale: float,
              want_features: bool):
        # ln_post on the class token only, then @ proj (clip/model.py:271-274) and - when text features are given - the
        # normalised, scaled cosine logits (model.py:399-404): ONE launch, both contractions on the matrix cores
        # (leclip_image_tail_fwd).  The class rows sit at a fixed stride (T*d) and are read in place.
        e = self.proj.shape[1]
        if self.width % 64 == 0 and self.width <= 1024 and e % 16 == 0:
            return ops.image_tail(x, batch, row_stride, self.ln_post_w, self.ln_post_b, self.proj_t, text_features, scale, want_features)
        feat = ops.gather_ln_proj(x, cls_rows, self.ln_post_w, self.ln_post_b, self.proj)     # odd widths (tiny test towers)
        return feat, (ops.l2norm_logits(feat, text_features, scale) if text_features is not None else None)

    def forward(self, image: torch.Tensor, taps: Optional[dict] = None, beside=None) -> torch.Tensor:
        """Image features [B, E] fp32 (VisionTransformer.forward).  ``beside``: see _on_streams."""
        if image.shape[0] == 0:   # an empty shard (ragged multi-GPU split): nothing to launch
            self._check_image(image)
            if beside is not None:
                self.beside_result = beside()
            return torch.empty((0, self.proj.shape[1]), dtype=torch.float32, device=self.device)

        def run(part):
            x, batch, cls_rows, stride = self._trunk(part, taps, cls_only=True)
            return (self._tail(x, batch, cls_rows, stride, None, 1.0, True)[0],)
        if taps is not None:
            if beside is not None:
                self.beside_result = beside()
            return run(image)[0]
        return self._on_streams(image, run, beside)[0]

    def dense_features(self, image: torch.Tensor) -> torch.Tensor:
        """[B, T, E] fp32: EVERY token of the last block through ln_post and proj (row 0 = the class token = forward()'s
        feature, rows 1.. = the patch tokens) - the per-position features of the local branch."""
        if image.shape[0] == 0:
            self._check_image(image)
            return torch.empty((0, self.tokens, self.proj.shape[1]), dtype=torch.float32, device=self.device)
        w = self.proj_t_padded()          # (built on the caller's stream, before any part is forked)

        def run(part):                    # large batches as stream parts, like forward() / score(): images are independent, same bits
            x, batch, _, _ = self._trunk(part)
            h = ops.layernorm(x, self.ln_post_w, self.ln_post_b)
            return (ops.gemm(h, w, out_dtype=torch.float32)[:, :self.proj.shape[1]].reshape(batch, self.tokens, -1),)
        return self._on_streams(image, run)[0]

    def proj_t_padded(self):
        """proj^T with its row count (E) rounded up to the GEMM kernels' N granularity (zero rows)."""
        if getattr(self, "_proj_t_pad", None) is None:
            e, d = self.proj_t.shape
            gran = 64 if self.dtype == torch.float32 else 128
            ep = (e + gran - 1) // gran * gran
            w = torch.zeros((ep, d), dtype=self.dtype, device=self.device)
            w[:e] = self.proj_t
            self._proj_t_pad = w
        return self._proj_t_pad

    def score(self, image: torch.Tensor, text_features: torch.Tensor, scale: float, want_features: bool = False):
        """(features or None, logits [B, C]): the image tower with the cosine-logit contraction folded into its tail kernel."""
        tf = text_features.float().contiguous()
        if image.shape[0] == 0:
            self._check_image(image)
            feat = torch.empty((0, self.proj.shape[1]), dtype=torch.float32, device=self.device) if want_features else None
            return feat, torch.empty((0, tf.shape[0]), dtype=torch.float32, device=self.device)

        def run(part):
            x, batch, cls_rows, stride = self._trunk(part, cls_only=True)
            return self._tail(x, batch, cls_rows, stride, tf, scale, want_features)
        return self._on_streams(image, run)


class TextEngine:
    """CLIP.encode_text / TextEncoder.forward (clip/model.py:379-392; trainers/Caption_distill_double.py:82-101)."""

    def __init__(self, clip_model, dtype: torch.dtype, device):
        self.dtype, self.device = dtype, device
        self.width = clip_model.ln_final.weight.shape[0]
        self.heads = self.width // 64
        self.table = _f32(clip_model.token_embedding.weight, device)
        self.pos = _f32(clip_model.positional_embedding, device)
        self.ln_w, self.ln_b = _f32(clip_model.ln_final.weight, device), _f32(clip_model.ln_final.bias, device)
        self.proj = clip_model.text_projection.detach().to(device=device, dtype=dtype).contiguous()
        self.proj_t = self.proj.t().contiguous()  # [E, d] nn.Linear layout for the if_sequence GEMM
        self.blocks = pack_blocks(clip_model.transformer.resblocks, dtype, device)
        self._ws: Dict[tuple, _Workspace] = {}

    def _workspace(self, n: int, t: int) -> _Workspace:
        # one workspace per (shape, HIP stream), as in VisionEngine: towers enqueued on different streams must not share activation buffers
        key = (n, t, torch.cuda.current_stream(self.device).cuda_stream if torch.cuda.is_available() else 0)
        if key not in self._ws:
            if len(self._ws) > 6:
                self._ws.clear()
            self._ws[key] = _Workspace(n * t, self.width, self.dtype, self.device)
        return self._ws[key]

    def _tower(self, x: torch.Tensor, n: int, t: int, taps: Optional[dict]):
        return run_blocks(x.view(n * t, self.width), self.blocks, self._workspace(n, t), n, t, self.heads, True, taps)

    def _pool(self, x: torch.Tensor, tokens: torch.Tensor, n: int, t: int, if_sequence: bool):
        if if_sequence:
            h = ops.layernorm(x, self.ln_w, self.ln_b)
            return ops.gemm(h, self.proj_t, out_dtype=torch.float32).view(n, t, -1)
        _, flat = ops.eot_index(tokens.to(self.device).contiguous())
        return ops.gather_ln_proj(x, flat, self.ln_w, self.ln_b, self.proj)

    def encode_tokens(self, tokens: torch.Tensor, if_sequence: bool = False, taps: Optional[dict] = None):
        tokens = tokens.to(self.device).contiguous()
        n, t = tokens.shape
        x = ops.embed_tokens(tokens, self.table, self.pos, self.dtype)
        x = self._tower(x, n, t, taps)
        return self._pool(x, tokens, n, t, if_sequence)

    def encode_prompts(self, prompts: torch.Tensor, tokenized_prompts: torch.Tensor, if_sequence: bool = False,
                       taps: Optional[dict] = None):
        n, t, _ = prompts.shape
        x = ops.add_pos(prompts.to(device=self.device, dtype=torch.float32), self.pos, self.dtype)
        x = self._tower(x, n, t, taps)
        return self._pool(x, tokenized_prompts, n, t, if_sequence)
