"""Tensor-level wrappers of the diffusion kernels of the C ABI (include/asd_hip.h: asd_gemm_f16, asd_groupnorm_f16,
asd_layernorm_f16, asd_softmax_f16, asd_softmax_bwd_f16, asd_geglu_f16, asd_silu_f16, asd_timestep_embedding_f16, asd_concat_f16, asd_attention_f16).
Activations are NHWC / token-major fp16 tensors; outputs are allocated with torch on the current stream."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from .. import _lib as L
from .._lib import GemmArgs, check, f32, i32, lib, ptr, stream

_zero = {}
_gn_stats = {}


def zero_page(device) -> torch.Tensor:
    key = str(device)
    if key not in _zero:
        _zero[key] = torch.zeros(256, dtype=torch.uint8, device=device)
    return _zero[key]


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


# ---- GEMM plans: (tile configuration, split-K) per problem shape ---------------------------------------------------
# The plan table and the autotuner live in the library (csrc/gemm.hip: asd_gemm_plan_*, asd_gemm_tune): asd_gemm_f16 with
# split_k = 0 uses the recorded plan of the shape, else its cost model.  The winners for the shapes of the shipped configs are
# committed in gemm_plans.json and pushed into the library when this module is imported; a shape met for the first time is tuned
# once (a few ms, never under graph capture) unless ASD_GEMM_AUTOTUNE=0.
TILE_BN = (64, 128, 64, 128, 320, 256, 320, 128, 64, 128, 64, 128, 64, 64, 128, 64, 64, 64, 64, 128, 128, 256, 320, 128, 160, 64, 64, 64, 128)
TILE_BM = (128, 128, 256, 256, 128, 256, 256, 320, 256, 256, 256, 256, 64, 256, 256, 64, 64, 64, 128, 128, 512, 256, 256, 256, 256, 320, 128, 64, 128)
WS_TILE = 25                             # weight-streaming 3x3 convolution of the 8x8 level (csrc/gemm_ws.hip): split_k >= 2, Cin % (32 split_k) == 0
WINDOW_TILES = (8, 9, 10, 11, 13, 14, 20, 21, 22, 23, 24)   # LDS-window 3x3 convolution (16x16-pixel patch x 64 / 128 channels); 10, 11: two blocks per CU; 13, 14: + four-wave form
PP_TILES = (20, 21, 22, 23, 24)          # ping-pong window convolution (csrc/gemm_pp.hip): whole N tiles, image rows % (TILE_BM / 16) == 0
# 15: 64x64 with a 4-stage operand ring; 16-19: intra-block split-K (64x64 x 2 / x 4 k-groups, 128x64 x 2, 128x128 x 2) — few-block launches
AUTOTUNE = os.environ.get("ASD_GEMM_AUTOTUNE", "1") != "0"
PLAN_FILE = os.environ.get("ASD_GEMM_PLAN_FILE", os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_plans.json"))
TUNE_SCRATCH_BYTES = 512 << 20
_tune_scratch = {}


def _sig(key):
    """(M, N, K, lda) | (M, N, K, (Hin, Cin, stride, upsample, pad)) -> the library's (M, N, K, conv, s0..s4)"""
    M, N, K, tail = key
    if isinstance(tail, tuple):
        return (M, N, K, 1) + tuple(int(v) for v in tail)
    return (M, N, K, 0, int(tail), 0, 0, 0, 0)


def load_plans(path: str = PLAN_FILE) -> int:
    import ast
    import json

    if not os.path.exists(path):
        return 0
    with open(path) as f:
        table = json.load(f)
    for k, v in table.items():
        check(lib().asd_gemm_plan_set(*[i32(x) for x in _sig(ast.literal_eval(k))], i32(int(v[0])), i32(int(v[1]))))
    return len(table)


def plan_table() -> dict:
    """the library's current plan table in the JSON file's key form"""
    out = {}
    buf = (C.c_int32 * 11)()
    for i in range(lib().asd_gemm_plan_count()):
        check(lib().asd_gemm_plan_entry(i32(i), buf))
        M, N, K, conv, s0, s1, s2, s3, s4, tile, sk = list(buf)
        out[(M, N, K, (s0, s1, s2, s3, s4) if conv else s0)] = (tile, sk)
    return out


def save_plans(path: str = PLAN_FILE) -> None:
    import json

    with open(path, "w") as f:
        json.dump({repr(k): list(v) for k, v in sorted(plan_table().items(), key=lambda kv: repr(kv[0]))}, f, indent=0)


def plan_of(g: GemmArgs):
    t, sk = C.c_int32(0), C.c_int32(1)
    tuned = lib().asd_gemm_plan_get(C.byref(g), C.byref(t), C.byref(sk)) == 0
    return (t.value, sk.value) if tuned else None


def default_split(M: int, N: int, K: int) -> int:
    """split-K of the library for an un-tuned plain GEMM of this shape (tools)"""
    g = GemmArgs()
    g.M, g.N, g.K, g.lda = M, N, K, K
    t, sk = C.c_int32(0), C.c_int32(1)
    lib().asd_gemm_plan_get(C.byref(g), C.byref(t), C.byref(sk))
    return sk.value


def tune_scratch(device) -> torch.Tensor:
    key = str(device)
    if key not in _tune_scratch:
        _tune_scratch[key] = torch.empty(TUNE_SCRATCH_BYTES, dtype=torch.uint8, device=device)
    return _tune_scratch[key]


def release_tune_scratch() -> None:
    _tune_scratch.clear()


if os.environ.get("ASD_GEMM_PLAN_FILE", "") != "none" and os.path.exists(L.LIB_PATH):
    load_plans()


def gemm(a: torch.Tensor, w: torch.Tensor, bias=None, row_bias=None, rows_per_group: int = 0, residual=None, act: int = 0,
         out: Optional[torch.Tensor] = None, out_f32: bool = False, split_k: Optional[int] = None, conv: Optional[dict] = None,
         M: Optional[int] = None, tile_cfg: int = 0, gn_rows: int = 0, gn_bwd: Optional[dict] = None, ln: Optional[dict] = None,
         gn_apply: Optional[dict] = None):
    """C = act(A W^T + bias + row_bias) + residual.  a: [M, K] fp16 (last dim contiguous) or NHWC image when conv.
    gn_rows > 0: also ask the epilogue for the GroupNorm statistics records of C (rows per batch element = gn_rows); returns
    (C, records | None, records_per_batch_element).  gn_bwd = dict(x, fstats, gamma, beta, eps, silu): C is the gradient reaching
    GroupNorm(x)[+SiLU] and the records carry that layer's two backward reductions instead (asd_gemm_args.gn_bwd_x).
    ln = dict(mode, sc, stats=None, eps=1e-5): a LayerNorm folded into this GEMM (asd_gemm_args.ln_mode; w = gamma (.) W and sc = fp32
    [2, rows] {rowsum(w), W beta} from weights._ln_fold): mode 1 normalises the rows of a (stats, if given, receives {mean, rstd} per
    row), mode 2 the rows of w with the statistics read from stats.
    gn_apply = dict(gamma, beta, eps, silu) with gn_rows: ask the PRODUCER to apply GroupNorm(32)(+SiLU) to C (asd_gemm_args.gn_apply:
    split-K launches whose reduction kernel owns whole groups); returns (C, y | None, stats | None) — y is None when this launch / plan
    cannot do it and the caller runs its own GroupNorm."""
    dev = a.device
    N, K = w.shape
    if conv is not None and int(conv.get("upsample", 0)) == 3:     # parity form: w = [4 parities][Cout][4 * Cin]
        N //= 4
    if conv is None:
        assert a.stride(-1) == 1 and a.dim() == 2
        M = a.shape[0]
        lda = a.stride(0)
    else:
        lda = 0
    if out is None:
        out = torch.empty((M, N // 2 if act == 2 else N), device=dev, dtype=torch.float32 if out_f32 else torch.float16)
    g = GemmArgs()
    g.A, g.W, g.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldw, g.ldc = lda, w.stride(0), out.stride(0)
    g.bias = None if bias is None else bias.data_ptr()
    g.row_bias = None if row_bias is None else row_bias.data_ptr()
    if row_bias is not None:     # a column slice of a wider matrix is used in place (row stride = its leading dimension)
        assert row_bias.stride(-1) == 1 and row_bias.shape[-1] == N, "row_bias: [groups, N] with unit column stride"
        g.ld_row_bias = row_bias.stride(0) if row_bias.dim() == 2 and row_bias.shape[0] > 1 else N
    g.rows_per_group = rows_per_group if row_bias is not None else 1
    g.residual = None if residual is None else residual.data_ptr()
    g.ldr = 0 if residual is None else residual.stride(0)
    g.act, g.out_f32 = act, int(out_f32)
    if conv is not None:
        g.conv = 1
        for k in ("Hin", "Win", "Cin", "Hout", "Wout", "stride", "pad", "upsample"):
            setattr(g, k, int(conv[k]))
    g.zero_page = zero_page(dev).data_ptr()
    g.tile_cfg = tile_cfg
    if ln is not None:
        g.ln_mode, g.ln_eps, g.ln_sc = int(ln["mode"]), float(ln.get("eps", 1e-5)), ln["sc"].data_ptr()
        g.ln_stats = None if ln.get("stats") is None else ln["stats"].data_ptr()
    if split_k is not None:
        g.split_k = split_k
    else:
        g.split_k = 0       # auto
        if AUTOTUNE and plan_of(g) is None and not torch.cuda.is_current_stream_capturing():
            sc = tune_scratch(dev)
            check(lib().asd_gemm_tune(C.byref(g), C.c_void_p(sc.data_ptr()), C.c_int64(sc.numel()), stream()))
    need = lib().asd_gemm_workspace_bytes(C.byref(g))
    ws = None
    if need > 0:
        ws = torch.empty(need, device=dev, dtype=torch.uint8)
        g.workspace = ws.data_ptr()
    rec, nrec = None, 0
    if gn_rows > 0 and gn_apply is not None:
        g.gn_cg, g.gn_rows, g.gn_apply = N // 32, gn_rows, 1
        y = st = None
        if lib().asd_gemm_gn_applies(C.byref(g)):
            y = torch.empty((M, N), device=dev, dtype=torch.float16)
            st = torch.empty((M // gn_rows, 64), device=dev, dtype=torch.float32)
            g.gn_apply_y, g.gn_apply_stats = y.data_ptr(), st.data_ptr()
            g.gn_apply_gamma, g.gn_apply_beta = gn_apply["gamma"].data_ptr(), gn_apply["beta"].data_ptr()
            g.gn_apply_eps, g.gn_apply_silu = float(gn_apply["eps"]), int(gn_apply["silu"])
        check(lib().asd_gemm_f16(C.byref(g), stream()))
        return out, y, st
    if gn_rows > 0:
        g.gn_cg, g.gn_rows = N // 32, gn_rows
        if gn_bwd is not None:
            g.gn_bwd_x, g.gn_bwd_fstats = gn_bwd["x"].data_ptr(), gn_bwd["fstats"].data_ptr()
            g.gn_bwd_gamma, g.gn_bwd_beta = gn_bwd["gamma"].data_ptr(), gn_bwd["beta"].data_ptr()
            g.gn_eps, g.gn_silu = float(gn_bwd["eps"]), int(gn_bwd["silu"])
        nrec = lib().asd_gemm_gn_records(C.byref(g))
        if nrec > 0:
            rec = torch.empty((M // gn_rows, nrec, 64), device=dev, dtype=torch.float32)
            g.gn_partials = rec.data_ptr()
    check(lib().asd_gemm_f16(C.byref(g), stream()))
    return (out, rec, nrec) if gn_rows > 0 else out


def conv3x3(x: torch.Tensor, w_packed: torch.Tensor, bias=None, stride: int = 1, pad: int = 1, upsample: bool = False,
            out_hw=None, **kw) -> torch.Tensor:
    """x: NHWC fp16 [B,H,W,Cin]; w_packed: [Cout, 9*Cin] with k = (ky, kx, cin). Returns [B,Ho,Wo,Cout].
    upsample: 0 plain, 1 fused nearest-2x, 2 input gradient of a stride-2 conv, 3 fused nearest-2x in its parity form
    (w_packed = pack_upsample_conv3x3_weight(w): [4 * Cout, 4 * Cin], four 2x2 convolutions over the low-resolution image)."""
    B, H, W_, Cin = x.shape
    assert x.is_contiguous()
    if out_hw is None:
        if upsample == 2:
            raise ValueError("transposed-conv mode needs out_hw")
        if upsample:
            Ho, Wo = 2 * H, 2 * W_
        else:
            Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W_ + 2 * pad - 3) // stride + 1
    else:
        Ho, Wo = out_hw
    conv = dict(Hin=H, Win=W_, Cin=Cin, Hout=Ho, Wout=Wo, stride=stride, pad=pad, upsample=int(upsample))  # upsample: 0/1/2
    y = gemm(x, w_packed, bias=bias, conv=conv, M=B * Ho * Wo, **kw)
    cout = w_packed.shape[0] // 4 if int(upsample) == 3 else w_packed.shape[0]
    if isinstance(y, tuple):      # gn_rows: (C, records, records per batch element)
        return y[0].view(B, Ho, Wo, cout), y[1], y[2]
    return y.view(B, Ho, Wo, cout)


def pack_upsample_conv3x3_weight(w: torch.Tensor) -> torch.Tensor:
    from .weights import _pack_upsample_conv3x3

    return _pack_upsample_conv3x3(w).contiguous()


def pack_conv3x3_weight(w: torch.Tensor, cin_pad: Optional[int] = None) -> torch.Tensor:
    from .weights import _pack_conv3x3

    return _pack_conv3x3(w, cin_pad).contiguous()


def groupnorm(x1: torch.Tensor, gamma, beta, eps: float, silu: bool, x2: Optional[torch.Tensor] = None,
              return_stats: bool = False):
    """x: [B, HW, C] (or [B,H,W,C]) NHWC fp16, 32 groups; x2 = second tensor of a channel concat."""
    B = x1.shape[0]
    c1 = x1.shape[-1]
    c2 = 0 if x2 is None else x2.shape[-1]
    hw = x1.numel() // (B * c1)
    y = torch.empty(tuple(x1.shape[:-1]) + (c1 + c2,), device=x1.device, dtype=torch.float16)
    stats = torch.empty(B * 64 + (512 + B) * 64, device=x1.device, dtype=torch.float32)   # ASD_GN_STATS_FLOATS(B)
    check(lib().asd_groupnorm_f16(ptr(x1), i32(c1), _p(x2), i32(c2), i32(B), i32(hw), ptr(gamma), ptr(beta), f32(eps),
                                  i32(int(silu)), ptr(y), ptr(stats), stream()))
    return (y, stats[:B * 64]) if return_stats else y


def groupnorm_apply(x: torch.Tensor, gamma, beta, eps: float, silu: bool, records: torch.Tensor):
    """GroupNorm of x [B, HW, C] whose producer left its statistics as records [B, n, 64] (gemm(..., gn_rows=HW))"""
    B, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (B * c)
    y = torch.empty_like(x)
    stats = torch.empty(B * 64 + (512 + B) * 64, device=x.device, dtype=torch.float32)   # ASD_GN_STATS_FLOATS(B)
    check(lib().asd_groupnorm_apply_f16(ptr(x), i32(c), i32(B), i32(hw), ptr(gamma), ptr(beta), f32(eps), i32(int(silu)), ptr(records),
                                        i32(records.shape[1]), ptr(y), ptr(stats), stream()))
    return y, stats[:B * 64]


def groupnorm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma, beta, eps: float, silu: bool, stats: torch.Tensor,
                  dx_add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """input gradient of GroupNorm(+SiLU); dx_add (same shape as x) is added to it in the same pass."""
    B, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (B * c)
    dx = torch.empty_like(x)
    bstats = torch.empty((512 + B) * 64, device=x.device, dtype=torch.float32)
    if dx_add is not None:
        assert dx_add.is_contiguous() and dx_add.numel() == x.numel() and dx_add.dtype == x.dtype
    check(lib().asd_groupnorm_bwd_f16(ptr(x), ptr(dy), i32(c), i32(B), i32(hw), ptr(gamma), ptr(beta), f32(eps), i32(int(silu)),
                                      ptr(stats), ptr(dx_add), ptr(dx), ptr(bstats), stream()))
    return dx


def groupnorm_bwd_apply(x: torch.Tensor, dy: torch.Tensor, gamma, beta, eps: float, silu: bool, stats: torch.Tensor, records: torch.Tensor,
                        dx_add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """groupnorm_bwd whose two reductions were left behind by dy's producer (gemm(..., gn_rows=HW, gn_bwd=...))"""
    B, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (B * c)
    dx = torch.empty_like(x)
    bstats = torch.empty((512 + B) * 64, device=x.device, dtype=torch.float32)
    check(lib().asd_groupnorm_bwd_apply_f16(ptr(x), ptr(dy), i32(c), i32(B), i32(hw), ptr(gamma), ptr(beta), f32(eps), i32(int(silu)),
                                            ptr(stats), ptr(records), i32(records.shape[1]), ptr(dx_add), ptr(dx), ptr(bstats), stream()))
    return dx


def transpose(x: torch.Tensor) -> torch.Tensor:
    rows, cols = x.shape
    y = torch.empty((cols, rows), device=x.device, dtype=torch.float16)
    check(lib().asd_transpose_f16(C.c_void_p(x.data_ptr()), i32(rows), i32(cols), i32(x.stride(0)), ptr(y), i32(rows), stream()))
    return y


def softmax(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """row softmax of scale * x, x [rows, cols] fp16 (row stride = x.stride(0))."""
    rows, cols = x.shape
    y = torch.empty((rows, cols), device=x.device, dtype=torch.float16)
    check(lib().asd_softmax_f16(C.c_void_p(x.data_ptr()), i32(x.stride(0)), i32(rows), i32(cols), f32(scale), ptr(y), i32(cols), stream()))
    return y


def softmax_bwd(p: torch.Tensor, dp: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """ds = scale * p o (dp - rowsum(dp o p)) for contiguous [rows, cols] fp16 tensors."""
    rows, cols = p.shape
    assert p.is_contiguous() and dp.is_contiguous()
    ds = torch.empty_like(p)
    check(lib().asd_softmax_bwd_f16(ptr(p), ptr(dp), i32(cols), i32(rows), i32(cols), f32(scale), ptr(ds), stream()))
    return ds


def layernorm(x: torch.Tensor, gamma, beta, eps: float = 1e-5) -> torch.Tensor:
    c = x.shape[-1]
    y = torch.empty_like(x)
    check(lib().asd_layernorm_f16(ptr(x), i32(x.numel() // c), i32(c), ptr(gamma), ptr(beta), f32(eps), ptr(y), stream()))
    return y


def pack_geglu_weight(w: torch.Tensor, b: torch.Tensor):
    """GEGLU.proj [2C', K] (value rows, then gate rows; attention.py:49-56) -> rows interleaved in 32-row groups
    [16 value | 16 gate] for the fused epilogue (asd_gemm_args.act = 2); the bias is permuted the same way."""
    from .weights import _geglu_rows

    perm = _geglu_rows(w.shape[0] // 2).to(w.device)
    return w[perm].contiguous(), b[perm].contiguous()


def geglu(h: torch.Tensor) -> torch.Tensor:
    c = h.shape[-1] // 2
    y = torch.empty(tuple(h.shape[:-1]) + (c,), device=h.device, dtype=torch.float16)
    check(lib().asd_geglu_f16(ptr(h), i32(h.numel() // (2 * c)), i32(c), ptr(y), stream()))
    return y


def silu(x: torch.Tensor) -> torch.Tensor:
    y = torch.empty_like(x)
    check(lib().asd_silu_f16(ptr(x), C.c_int64(x.numel()), ptr(y), stream()))
    return y


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    t = t.float().contiguous()
    y = torch.empty((t.shape[0], dim), device=t.device, dtype=torch.float16)
    check(lib().asd_timestep_embedding_f16(ptr(t), i32(t.shape[0]), i32(dim), ptr(y), stream()))
    return y


def concat(x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    c1, c2 = x1.shape[-1], x2.shape[-1]
    y = torch.empty(tuple(x1.shape[:-1]) + (c1 + c2,), device=x1.device, dtype=torch.float16)
    check(lib().asd_concat_f16(ptr(x1), i32(c1), ptr(x2), i32(c2), C.c_int64(x1.numel() // c1), ptr(y), stream()))
    return y


def attention(q: torch.Tensor, k: torch.Tensor, vT: torch.Tensor, batch: int, heads: int, lq: int, lk: int,
              lk_stride: Optional[int] = None, scale: Optional[float] = None) -> torch.Tensor:
    """q: [batch*lq, >=heads*64] (row stride ldq), k: [batch*lk_stride, ...], vT: [heads*64, batch*lk_stride]."""
    lk_stride = lk if lk_stride is None else lk_stride
    o = torch.empty((batch * lq, heads * 64), device=q.device, dtype=torch.float16)
    check(lib().asd_attention_f16(C.c_void_p(q.data_ptr()), i32(q.stride(0)), C.c_void_p(k.data_ptr()), i32(k.stride(0)),
                                  C.c_void_p(vT.data_ptr()), i32(vT.stride(0)), ptr(o), i32(o.stride(0)), i32(batch), i32(heads),
                                  i32(lq), i32(lk), i32(lk_stride), f32(scale if scale is not None else 64 ** -0.5),
                                  ptr(zero_page(q.device)), stream()))
    return o
