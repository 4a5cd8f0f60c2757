"""ctypes binding of libfairygen_hip.so (C ABI: include/fairygen_hip.h).

This is the ONLY compute backend of the package.  There is no CPU or eager-PyTorch fallback: if the
library is missing, or a tensor is not a contiguous bf16 tensor on a HIP device, the call raises.
PyTorch is used for device memory, streams and the plain GEMMs (hipBLASLt) only.
"""
import ctypes
import os

import torch

# FAIRYGEN_HIP_LIB: load an alternative build of the same ABI (kernel A/B experiments); still no fallback.
_LIB_PATH = os.environ.get("FAIRYGEN_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfairygen_hip.so")
_lib = None

ABI_VERSION = 3

_i64, _i32, _f32, _vp = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p

# name -> argtypes; every function returns int except where noted.  Mirrors include/fairygen_hip.h.
_SIGNATURES = {
    "fg_ln_modulate_bf16": [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _i64, _i64, _i64, _vp],
    "fg_ln_affine_bf16": [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp],
    "fg_ln_modulate_fp8_bf16": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _i64, _i64, _i64, _f32, _vp],
    "fg_residual_ln_fp8_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _f32, _i64, _i64, _i64, _f32, _vp],
    "fg_gate_residual_bf16": [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _i64, _i64, _vp],
    "fg_residual_ln_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _f32, _i64, _i64, _i64, _vp],
    "fg_rmsnorm_rope_bf16": [_vp, _i64, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _f32, _vp],
    "fg_rmsnorm_rope_grouped_bf16": [_vp, _i64, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _f32, _i32, _i64, _i64, _vp],
    "fg_copy_groups_bf16": [_vp, _i64, _i64, _vp, _i64, _i64, _i32, _i64, _i32, _vp],
    "fg_fp8_quant_rows_bf16": [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp],
    "fg_act_bf16": [_vp, _vp, _i64, _i32, _vp],
    "fg_gemm_epilogue_bf16": [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _i64, _i64, _i64, _vp, _vp],
    "fg_gemm_fp8_bf16": [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _i64, _i64, _i64, _vp, _vp],
    "fg_attn_fwd_bf16": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _i64, _i64, _i32, _i32, _f32, _vp, _i64, _vp],
    "fg_cfg_euler_bf16": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _vp],
    "fg_vae_rmsnorm_silu_bf16": [_vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "fg_conv_pack_weight_bf16": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "fg_conv3d_cl_bf16": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "fg_dupup3d_add_bf16": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "fg_softmax_rows_f32_bf16": [_vp, _vp, _i64, _i64, _f32, _vp],
    "fg_vae_latent_to_cl_bf16": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "fg_vae_unpatchify_bf16": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "fg_vae_tile_accumulate_bf16": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "fg_vae_tile_finalize_bf16": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "fg_vae_patchify_bf16": [_vp, _vp, _i32, _i32, _i32, _vp],
    "fg_avgdown3d_add_bf16": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "fg_vae_latent_from_cl_bf16": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "fg_video_to_uint8": [_vp, _vp, _i32, _i32, _i32, _vp],
    "fg_softmax_bias_bf16": [_vp, _vp, _vp, _vp, _i64, _i64, _vp],
    "fg_gated_gelu_bf16": [_vp, _vp, _vp, _i64, _vp],
}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ["fg_version", "fg_last_error", "fg_conv_packed_bytes", "fg_attn_workspace_bytes", "fg_attn_split_choice",
                                                   "fg_conv_tile_choice", "fg_gemm_workspace_bytes", "fg_gemm_debug_grid"])


class HipLibraryError(RuntimeError):
    pass


def library_path():
    return _LIB_PATH


def load():
    """Load libfairygen_hip.so (built by __graft_entry__.build() / make -C fairygen_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipLibraryError(
            f"{_LIB_PATH} not found: build it with `make -C fairygen_amd/csrc` (or __graft_entry__.build()). "
            "fairygen_amd has no CPU / PyTorch fallback for its kernels.")
    lib = ctypes.CDLL(_LIB_PATH)
    lib.fg_version.restype = ctypes.c_int
    lib.fg_last_error.restype = ctypes.c_char_p
    lib.fg_conv_packed_bytes.restype = ctypes.c_int64
    lib.fg_conv_packed_bytes.argtypes = [_i32] * 5
    lib.fg_attn_workspace_bytes.restype = ctypes.c_int64
    lib.fg_attn_workspace_bytes.argtypes = [_i32, _i64, _i64, _i32]
    lib.fg_conv_tile_choice.restype = ctypes.c_int
    lib.fg_conv_tile_choice.argtypes = [_i32] * 4
    lib.fg_gemm_workspace_bytes.restype = ctypes.c_int64
    lib.fg_gemm_workspace_bytes.argtypes = [_i64] * 3
    lib.fg_gemm_debug_grid.restype = ctypes.c_int
    lib.fg_gemm_debug_grid.argtypes = [_i32]
    lib.fg_attn_split_choice.restype = ctypes.c_int
    lib.fg_attn_split_choice.argtypes = [_i32, _i64, _i64, _i32, _i64, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = argtypes
    if lib.fg_version() != ABI_VERSION:
        raise HipLibraryError(f"ABI mismatch: library {lib.fg_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def _call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise HipLibraryError(f"{name} failed ({rc}): {lib.fg_last_error().decode()}")


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _dev(t, name, dtype=torch.bfloat16):
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise HipLibraryError(f"{name}: expected a tensor on a HIP device, got {getattr(t, 'device', type(t))} "
                              "(no CPU fallback)")
    if t.dtype != dtype:
        raise HipLibraryError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _rows(t, name):
    """View (..., C) contiguous as (rows, C)."""
    _dev(t, name)
    if not t.is_contiguous():
        raise HipLibraryError(f"{name}: must be contiguous")
    return t.numel() // t.shape[-1], t.shape[-1]


class ModTable:
    """AdaLN modulation rows: tensor (mod_rows, K, C) bf16; vector j of row r = table[r, j].

    mod_rows 1 (T2V), 2 (TI2V: tokens < first_rows use row 0) or N (per token)."""

    def __init__(self, table, first_rows=0):
        _dev(table, "mod table")
        assert table.dim() == 3 and table.is_contiguous()
        self.table, self.first_rows = table, int(first_rows)
        self.mod_rows, self.k, self.c = table.shape
        self.ld = self.k * self.c

    def vec(self, j):
        return ctypes.c_void_p(self.table.data_ptr() + 2 * j * self.c)


def _mod_args(mod, rows):
    if mod.mod_rows not in (1, 2, rows):
        raise HipLibraryError(f"modulation table has {mod.mod_rows} rows; expected 1, 2 or {rows}")
    return mod.mod_rows, mod.first_rows, mod.ld


# ------------------------------------------------------------------------------------- DiT kernels
def ln_modulate(x, mod, shift_idx, scale_idx, eps, out=None):
    rows, c = _rows(x, "x")
    out = torch.empty_like(x) if out is None else out
    _call("fg_ln_modulate_bf16", _ptr(x), mod.vec(shift_idx), mod.vec(scale_idx), _ptr(out), rows, c, eps,
          *_mod_args(mod, rows), _stream(x))
    return out


def ln_affine(x, w, b, eps, out=None):
    rows, c = _rows(x, "x")
    _dev(w, "w"), _dev(b, "b")
    out = torch.empty_like(x) if out is None else out
    _call("fg_ln_affine_bf16", _ptr(x), _ptr(w), _ptr(b), _ptr(out), rows, c, eps, _stream(x))
    return out


def gate_residual(x, y, mod=None, gate_idx=None, out=None):
    rows, c = _rows(x, "x")
    _rows(y, "y")
    out = torch.empty_like(x) if out is None else out
    if mod is None:
        _call("fg_gate_residual_bf16", _ptr(x), _ptr(y), None, _ptr(out), rows, c, 1, 0, 0, _stream(x))
    else:
        _call("fg_gate_residual_bf16", _ptr(x), _ptr(y), mod.vec(gate_idx), _ptr(out), rows, c,
              *_mod_args(mod, rows), _stream(x))
    return out


def residual_ln_modulate(x, y, mod, gate_idx, shift_idx, scale_idx, eps, x_out=None, norm_out=None, norm_mod=None):
    """x_out = x + gate*y (gate_idx None: x + y); norm_out = modulate(LN(x_out)).

    gate comes from `mod`, shift/scale from `norm_mod` (default: `mod`); both tables must have the same
    number of rows, first_rows and leading dimension."""
    rows, c = _rows(x, "x")
    _rows(y, "y")
    norm_mod = mod if norm_mod is None else norm_mod
    if (norm_mod.mod_rows, norm_mod.first_rows, norm_mod.ld) != (mod.mod_rows, mod.first_rows, mod.ld):
        raise HipLibraryError("residual_ln_modulate: gate and norm tables must share rows / first_rows / ld")
    x_out = torch.empty_like(x) if x_out is None else x_out
    norm_out = torch.empty_like(x) if norm_out is None else norm_out
    gate = mod.vec(gate_idx) if gate_idx is not None else None
    _call("fg_residual_ln_bf16", _ptr(x), _ptr(y), gate, _ptr(x_out), norm_mod.vec(shift_idx),
          norm_mod.vec(scale_idx), _ptr(norm_out), 0, rows, c, eps, *_mod_args(mod, rows), _stream(x))
    return x_out, norm_out


def residual_ln_affine(x, y, w, b, eps, mod=None, gate_idx=None, x_out=None, norm_out=None):
    """x_out = x + gate*y (mod None: x + y); norm_out = LN(x_out)*w + b."""
    rows, c = _rows(x, "x")
    _rows(y, "y")
    _dev(w, "w"), _dev(b, "b")
    x_out = torch.empty_like(x) if x_out is None else x_out
    norm_out = torch.empty_like(x) if norm_out is None else norm_out
    if mod is None:
        gate, margs = None, (1, 0, 0)
    else:
        gate, margs = mod.vec(gate_idx), _mod_args(mod, rows)
    _call("fg_residual_ln_bf16", _ptr(x), _ptr(y), gate, _ptr(x_out), _ptr(w), _ptr(b), _ptr(norm_out), 1, rows, c,
          eps, *margs, _stream(x))
    return x_out, norm_out


def _fp8_rows_out(x):
    rows, c = _rows(x, "x")
    return (torch.empty((rows, c), dtype=torch.float8_e4m3fn, device=x.device),
            torch.empty((rows, 1), dtype=torch.float32, device=x.device))


def ln_modulate_fp8(x, mod, shift_idx, scale_idx, eps):
    """ln_modulate whose result goes to an fp8 Linear only: (x_fp8 (rows, C) e4m3fn, scale (rows, 1) fp32) = what fp8_quant_rows
    makes of ln_modulate(x, ...), without the bf16 row in HBM."""
    rows, c = _rows(x, "x")
    q, sc = _fp8_rows_out(x)
    _call("fg_ln_modulate_fp8_bf16", _ptr(x), mod.vec(shift_idx), mod.vec(scale_idx), _ptr(q), _ptr(sc), rows, c, eps,
          *_mod_args(mod, rows), FP8_E4M3FN_MAX, _stream(x))
    return q, sc


def residual_ln_modulate_fp8(x, y, mod, gate_idx, shift_idx, scale_idx, eps, x_out=None, norm_mod=None):
    """residual_ln_modulate with the normalised row as (fp8 rows, scales): returns x_out, (x_fp8, scale)."""
    rows, c = _rows(x, "x")
    _rows(y, "y")
    norm_mod = mod if norm_mod is None else norm_mod
    if (norm_mod.mod_rows, norm_mod.first_rows, norm_mod.ld) != (mod.mod_rows, mod.first_rows, mod.ld):
        raise HipLibraryError("residual_ln_modulate_fp8: gate and norm tables must share rows / first_rows / ld")
    x_out = torch.empty_like(x) if x_out is None else x_out
    q, sc = _fp8_rows_out(x)
    gate = mod.vec(gate_idx) if gate_idx is not None else None
    _call("fg_residual_ln_fp8_bf16", _ptr(x), _ptr(y), gate, _ptr(x_out), norm_mod.vec(shift_idx), norm_mod.vec(scale_idx),
          _ptr(q), _ptr(sc), 0, rows, c, eps, *_mod_args(mod, rows), FP8_E4M3FN_MAX, _stream(x))
    return x_out, (q, sc)


def residual_ln_affine_fp8(x, y, w, b, eps, mod=None, gate_idx=None, x_out=None):
    """residual_ln_affine with the normalised row as (fp8 rows, scales): returns x_out, (x_fp8, scale)."""
    rows, c = _rows(x, "x")
    _rows(y, "y")
    _dev(w, "w"), _dev(b, "b")
    x_out = torch.empty_like(x) if x_out is None else x_out
    q, sc = _fp8_rows_out(x)
    if mod is None:
        gate, margs = None, (1, 0, 0)
    else:
        gate, margs = mod.vec(gate_idx), _mod_args(mod, rows)
    _call("fg_residual_ln_fp8_bf16", _ptr(x), _ptr(y), gate, _ptr(x_out), _ptr(w), _ptr(b), _ptr(q), _ptr(sc), 1, rows, c,
          eps, *margs, FP8_E4M3FN_MAX, _stream(x))
    return x_out, (q, sc)


def rmsnorm_rope(x, weight, num_heads, eps, cos=None, sin=None, out=None, grouped=None):
    """x: (..., C) possibly a column slice of a wider row-major buffer (stride(-2) = ld).

    RoPE tables: cos and sin fp64 (rows, head_dim/2) -> the reference's fp64 rotation; or cos = ONE fp32 interleaved
    (rows, head_dim/2, 2) table {cos, sin} with sin=None -> fp32 FMA rotation (the fast default of the pipeline).

    grouped=(dst, group_cols, group_stride, ld): write column block g of row r to the 1-D view
    dst[g*group_stride + r*ld : ... + group_cols] instead of a (rows, C) tensor (Ulysses send buffer); returns dst."""
    _dev(x, "x"), _dev(weight, "weight")
    c = x.shape[-1]
    if x.stride(-1) != 1:
        raise HipLibraryError("rmsnorm_rope: last dim must be dense")
    x2 = x.reshape(-1, c) if x.is_contiguous() else x
    if x2.dim() != 2:
        x2 = x2.squeeze(0)
    if x2.dim() != 2:
        raise HipLibraryError("rmsnorm_rope: strided input must be 2-D (rows, C) or (1, rows, C)")
    rows, ld = x2.shape[0], x2.stride(0)
    if grouped is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device) if out is None else out
    f32tab = 0
    if cos is not None and sin is None:
        f32tab = 1
        _dev(cos, "rope table", torch.float32)
        if cos.shape != (rows, c // num_heads // 2, 2) or not cos.is_contiguous():
            raise HipLibraryError(f"rmsnorm_rope: the fp32 rope table must be ({rows}, {c // num_heads // 2}, 2) contiguous")
    elif cos is not None:
        _dev(cos, "cos", torch.float64), _dev(sin, "sin", torch.float64)
        if cos.shape != (rows, c // num_heads // 2) or not cos.is_contiguous() or not sin.is_contiguous():
            raise HipLibraryError(f"rmsnorm_rope: rope tables must be ({rows}, {c // num_heads // 2}) contiguous")
    if grouped is not None:
        dst, group_cols, group_stride, out_ld = grouped
        _dev(dst, "grouped dst")
        groups = c // group_cols
        if dst.dim() != 1 or dst.stride(0) != 1 or \
                (groups - 1) * group_stride + max(rows - 1, 0) * out_ld + group_cols > dst.numel():
            raise HipLibraryError("rmsnorm_rope: grouped destination too small for the requested layout")
        _call("fg_rmsnorm_rope_grouped_bf16", _ptr(x2), ld, _ptr(weight), _ptr(cos), _ptr(sin), f32tab, _ptr(dst), rows, c,
              num_heads, eps, group_cols, group_stride, out_ld, _stream(x))
        return dst
    _call("fg_rmsnorm_rope_bf16", _ptr(x2), ld, _ptr(weight), _ptr(cos), _ptr(sin), f32tab, _ptr(out), rows, c, num_heads,
          eps, _stream(x))
    return out


def copy_groups(src, src_group_stride, src_ld, dst, dst_group_stride, dst_ld, groups, rows, cols):
    """dst[g*dst_group_stride + r*dst_ld + c] = src[g*src_group_stride + r*src_ld + c]; src / dst are 1-D views that
    start at the first element to move (strides in elements)."""
    _dev(src, "src"), _dev(dst, "dst")
    for t, gs, ld, name in ((src, src_group_stride, src_ld, "src"), (dst, dst_group_stride, dst_ld, "dst")):
        if t.dim() != 1 or t.stride(0) != 1 or (groups - 1) * gs + max(rows - 1, 0) * ld + cols > t.numel():
            raise HipLibraryError(f"copy_groups: {name} view too small for the requested layout")
    _call("fg_copy_groups_bf16", _ptr(src), src_group_stride, src_ld, _ptr(dst), dst_group_stride, dst_ld, groups, rows,
          cols, _stream(src))
    return dst


_gemm_workspace = {}      # (device, stream) -> scratch for the k-split pieces of a GEMM's last round (one fp32 tile per CU, reused)


def _gemm_ws(x, m, n, k_bytes, workspace):
    """The k-split scratch, only where fg_gemm_* would use it: at least 96 k-steps of 128 operand bytes, or at most 8 rounds of tiles per CU
    (launch_gemm in csrc/dit_gemm.hip; 256 CUs assumed here — a spare allocation on other devices, never a missing one)."""
    tiles = ((m + 255) // 256) * (n // 256)
    need = load().fg_gemm_workspace_bytes(m, n, k_bytes) if workspace and (k_bytes // 128 >= 96 or tiles <= 8 * 256) else 0
    if need <= 0:
        return None
    key = (x.device, torch.cuda.current_stream(x.device).cuda_stream)      # concurrent streams must not share scratch
    ws = _gemm_workspace.get(key)
    if ws is None or ws.numel() < need:
        ws = _gemm_workspace[key] = torch.empty(need, dtype=torch.uint8, device=x.device)
    return ws


def _gemm_out(name, x, m, n, out, residual):
    if out is None:
        if residual:
            raise HipLibraryError(f"{name}: residual=True needs the contiguous residual stream as `out`")
        return torch.empty(x.shape[:-1] + (n,), dtype=torch.bfloat16, device=x.device)
    _dev(out, "out")
    if not out.is_contiguous() or out.numel() != m * n:
        raise HipLibraryError(f"{name}: `out` must be a contiguous bf16 tensor of {m} x {n} elements")
    return out


def _gemm_mode(name, residual, mod, gate_idx, act, n):
    """(mode, gate pointer, gate rows, gate ld, first rows) of the fg_gemm_* epilogue."""
    if act not in (None, "gelu_tanh") or (act is not None and residual):
        raise HipLibraryError(f"{name}: act must be None or 'gelu_tanh', and not combined with residual=True")
    if residual and mod is not None:
        if mod.mod_rows not in (1, 2) or mod.c != n:
            raise HipLibraryError(f"{name}: the gate table must have 1 or 2 rows of N values")
        return 2, mod.vec(gate_idx), mod.mod_rows, mod.ld, mod.first_rows
    return (3 if residual else 4 if act else 0), None, 1, n, 0


def gemm_epilogue(x, weight, bias, out=None, residual=False, mod=None, gate_idx=None, workspace=True, act=None):
    """Linear on the persistent MFMA kernel.  residual=False: out = act(x @ weight^T + bias), act None or "gelu_tanh" (applied to the
    bf16-rounded Linear output and rounded again, like nn.Linear followed by nn.GELU).  residual=True: `out` holds the residual
    stream and becomes out + gate * (x @ weight^T + bias) (gate = vector gate_idx of `mod`, or 1 when mod is None), with the
    reference's rounding points (GateModule, models/wan_video_dit.py:188-193).  workspace=False: no k-split of the last round's tiles
    (every element one k-ordered accumulation, independent of the row count)."""
    _dev(x, "x"), _dev(weight, "weight"), _dev(bias, "bias")
    k = x.shape[-1]
    n = weight.shape[0]
    if weight.shape != (n, k) or not weight.is_contiguous() or bias.shape != (n,) or x.stride(-1) != 1:
        raise HipLibraryError("gemm_epilogue: weight must be a contiguous (N, K) tensor, bias (N,), x dense in its last dim")
    x2 = x.reshape(-1, k) if x.is_contiguous() else (x.squeeze(0) if x.dim() == 3 else x)
    if x2.dim() != 2:
        raise HipLibraryError("gemm_epilogue: strided input must be 2-D (rows, K) or (1, rows, K)")
    m, lda = x2.shape[0], x2.stride(0)
    out = _gemm_out("gemm_epilogue", x, m, n, out, residual)
    mode, gate, gate_rows, gate_ld, first = _gemm_mode("gemm_epilogue", residual, mod, gate_idx, act, n)
    ws = _gemm_ws(x, m, n, 2 * k, workspace)
    _call("fg_gemm_epilogue_bf16", _ptr(x2), lda, _ptr(weight), _ptr(bias), _ptr(out), n, m, n, k, mode, gate, gate_rows, gate_ld, first,
          _ptr(ws), _stream(x))
    return out


def gemm_fp8(x_fp8, scale_a, weight_fp8, bias, out=None, residual=False, mod=None, gate_idx=None, workspace=True, act=None, lead_shape=None):
    """torch._scaled_mm(x_fp8, weight_fp8.T, scale_a (rows, 1), ones (1, out), bias, out_dtype=bf16) of AutoWrappedLinear.fp8_linear
    (core/vram/layers.py:343-357) on the persistent kernel's e4m3 form, with the epilogues of gemm_epilogue.  x_fp8: (rows, K)
    float8_e4m3fn, scale_a: (rows, 1) fp32 (both from fp8_quant_rows / the fp8-output norm kernels), weight_fp8: (N, K) float8_e4m3fn.
    Returns (*lead_shape, N) bf16 (default lead_shape: (rows,))."""
    _dev(x_fp8, "x_fp8", torch.float8_e4m3fn), _dev(weight_fp8, "weight_fp8", torch.float8_e4m3fn), _dev(bias, "bias")
    _dev(scale_a, "scale_a", torch.float32)
    if x_fp8.dim() != 2 or x_fp8.stride(1) != 1 or not weight_fp8.is_contiguous() or weight_fp8.dim() != 2:
        raise HipLibraryError("gemm_fp8: x_fp8 must be (rows, K) with a dense last dim, weight_fp8 a contiguous (N, K) tensor")
    (m, k), n = x_fp8.shape, weight_fp8.shape[0]
    if weight_fp8.shape[1] != k or bias.shape != (n,) or scale_a.numel() != m or not scale_a.is_contiguous():
        raise HipLibraryError("gemm_fp8: shapes of weight_fp8 (N, K), bias (N,), scale_a (rows, 1) do not match x_fp8 (rows, K)")
    lead = (m,) if lead_shape is None else tuple(lead_shape)
    if out is None and not residual:
        out = torch.empty(lead + (n,), dtype=torch.bfloat16, device=x_fp8.device)
    out = _gemm_out("gemm_fp8", x_fp8, m, n, out, residual)
    mode, gate, gate_rows, gate_ld, first = _gemm_mode("gemm_fp8", residual, mod, gate_idx, act, n)
    ws = _gemm_ws(x_fp8, m, n, k, workspace)
    _call("fg_gemm_fp8_bf16", _ptr(x_fp8), x_fp8.stride(0), _ptr(scale_a), _ptr(weight_fp8), _ptr(bias), _ptr(out), n, m, n, k, mode, gate,
          gate_rows, gate_ld, first, _ptr(ws), _stream(x_fp8))
    return out


FP8_E4M3FN_MAX = 448.0


def fp8_quant_rows(x, act=None):
    """Per-row dynamic fp8 quantisation of AutoWrappedLinear.fp8_linear (core/vram/layers.py:331-342).
    x (..., C) bf16 with dense last dim (may be a column slice, 2-D/3-D) -> (x_fp8 (rows, C) float8_e4m3fn,
    scale_a (rows, 1) fp32).  act="gelu_tanh" applies the activation (rounded to bf16) before quantising."""
    _dev(x, "x")
    c = x.shape[-1]
    if x.stride(-1) != 1:
        raise HipLibraryError("fp8_quant_rows: last dim must be dense")
    x2 = x.reshape(-1, c) if x.is_contiguous() else (x.squeeze(0) if x.dim() == 3 else x)
    if x2.dim() != 2:
        raise HipLibraryError("fp8_quant_rows: strided input must be 2-D (rows, C) or (1, rows, C)")
    rows, ld = x2.shape[0], x2.stride(0)
    out = torch.empty((rows, c), dtype=torch.float8_e4m3fn, device=x.device)
    scale = torch.empty((rows, 1), dtype=torch.float32, device=x.device)
    _call("fg_fp8_quant_rows_bf16", _ptr(x2), ld, _ptr(out), _ptr(scale), None, rows, c,
          {None: 0, "gelu_tanh": 1}[act], FP8_E4M3FN_MAX, _stream(x))
    return out, scale


def activation(x, kind, out=None):
    _dev(x, "x")
    if not x.is_contiguous():
        raise HipLibraryError("activation: must be contiguous")
    out = x if out is None else out
    _call("fg_act_bf16", _ptr(x), _ptr(out), x.numel(), {"silu": 0, "gelu_tanh": 1}[kind], _stream(x))
    return out


def _ld_rows(t, name):
    """(B, N, HD) with dense last dim and B stride = N*ld."""
    _dev(t, name)
    if t.dim() != 3 or t.stride(2) != 1:
        raise HipLibraryError(f"{name}: expected (B, N, H*D) with dense last dim")
    ld = t.stride(1)
    if t.shape[0] > 1 and t.stride(0) != t.shape[1] * ld:
        raise HipLibraryError(f"{name}: batch stride must be N*ld")
    return ld


_attn_workspace = {}      # (device, stream) -> scratch tensor for the split-KV partials (grown on demand, reused)


def pow2_softmax_scale(head_dim):
    """(scale', f) with scale' * log2(e) = the power of two next to head_dim^-0.5 * log2(e) and f = their ratio (1.0201 for d = 128):
    softmax(scale' (f q) k^T) = softmax(q k^T / sqrt(d)), and fg_attn_fwd_bf16 runs its pre-multiplied form exactly for scale'."""
    import math
    sl = float(head_dim) ** -0.5 * math.log2(math.e)
    p = 2.0 ** round(math.log2(sl))
    return p / math.log2(math.e), sl / p


def attention(q, k, v, num_heads, out=None, scale=None):
    """softmax(scale q k^T) v with scale = 1 / sqrt(d) by default, "b s (n d)" in and out (AttentionModule semantics)."""
    ldq, ldk, ldv = _ld_rows(q, "q"), _ld_rows(k, "k"), _ld_rows(v, "v")
    b, nq, hd = q.shape
    nkv = k.shape[1]
    d = hd // num_heads
    out = torch.empty((b, nq, hd), dtype=q.dtype, device=q.device) if out is None else out
    need = load().fg_attn_workspace_bytes(b, nq, nkv, num_heads)
    key = (q.device, torch.cuda.current_stream(q.device).cuda_stream)      # concurrent streams must not share scratch
    ws = _attn_workspace.get(key)
    if need > 0 and (ws is None or ws.numel() < need):
        ws = _attn_workspace[key] = torch.empty(need, dtype=torch.uint8, device=q.device)
    _call("fg_attn_fwd_bf16", _ptr(q), ldq, _ptr(k), ldk, _ptr(v), ldv, _ptr(out), b, nq, nkv, num_heads, d,
          float(d) ** -0.5 if scale is None else float(scale), _ptr(ws) if need > 0 else None, need, _stream(q))
    return out


def cfg_euler(latents, posi, nega, cfg_scale, dsigma, out=None):
    _dev(latents, "latents"), _dev(posi, "posi")
    assert latents.is_contiguous() and posi.is_contiguous() and (nega is None or nega.is_contiguous())
    out = torch.empty_like(latents) if out is None else out
    _call("fg_cfg_euler_bf16", _ptr(latents), _ptr(posi), _ptr(nega), _ptr(out), latents.numel(), float(cfg_scale),
          float(dsigma), _stream(latents))
    return out


# ------------------------------------------------------------------------------------- VAE kernels
def vae_rmsnorm_silu(x, gamma, silu=True, out=None):
    pixels, c = _rows(x, "x")
    _dev(gamma, "gamma")
    out = torch.empty_like(x) if out is None else out
    _call("fg_vae_rmsnorm_silu_bf16", _ptr(x), _ptr(gamma), _ptr(out), pixels, c, int(silu), _stream(x))
    return out


def conv_pack_weight(w):
    """(Cout,Cin,kt,kh,kw) or (Cout,Cin,kh,kw) bf16 -> packed buffer for conv3d_cl."""
    _dev(w, "w")
    w = w.contiguous()
    if w.dim() == 4:
        w = w.unsqueeze(2)
    cout, cin, kt, kh, kw = w.shape
    nbytes = load().fg_conv_packed_bytes(cout, cin, kt, kh, kw)
    packed = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w.device)
    _call("fg_conv_pack_weight_bf16", _ptr(w), _ptr(packed), cout, cin, kt, kh, kw, _stream(w))
    return packed


def conv3d_cl(x, w_packed, bias, cout, kt, ks, residual=None, upsample2x=False, time_interleave=False, out=None,
              downsample2x=False):
    """x (T + kt-1, Hin, Win, Cin) channels-last, the first kt-1 frames being the causal history (feature cache)
    -> (T,H,W,Cout) [or (2T,H,W,Cout/2) with time_interleave]."""
    _dev(x, "x"), _dev(w_packed, "w_packed"), _dev(bias, "bias")
    assert x.dim() == 4 and x.is_contiguous() and x.shape[0] > kt - 1
    t, hin, win, cin = x.shape[0] - (kt - 1), x.shape[1], x.shape[2], x.shape[3]
    assert not (upsample2x and downsample2x)
    h, w = (hin * 2, win * 2) if upsample2x else ((hin // 2, win // 2) if downsample2x else (hin, win))
    oshape = (2 * t, h, w, cout // 2) if time_interleave else (t, h, w, cout)
    out = torch.empty(oshape, dtype=x.dtype, device=x.device) if out is None else out
    assert tuple(out.shape) == oshape and out.is_contiguous()
    if residual is not None:
        _dev(residual, "residual")
        assert tuple(residual.shape) == oshape and residual.is_contiguous()
    _call("fg_conv3d_cl_bf16", _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(residual), _ptr(out), t, h, w,
          cin, cout, kt, ks, 1 if upsample2x else (2 if downsample2x else 0), int(time_interleave), _stream(x))
    return out


def dupup3d_add(x, main, cout, ft, fs, first_chunk, out=None):
    _dev(x, "x"), _dev(main, "main")
    t, h, w, cin = x.shape
    oshape = (t * ft - (ft - 1 if first_chunk else 0), h * fs, w * fs, cout)
    assert tuple(main.shape) == oshape and main.is_contiguous() and x.is_contiguous()
    out = torch.empty_like(main) if out is None else out
    _call("fg_dupup3d_add_bf16", _ptr(x), _ptr(main), _ptr(out), t, h, w, cin, cout, ft, fs, int(first_chunk),
          _stream(x))
    return out


def softmax_rows(scores, scale):
    _dev(scores, "scores", torch.float32)
    assert scores.dim() == 2 and scores.is_contiguous()
    probs = torch.empty(scores.shape, dtype=torch.bfloat16, device=scores.device)
    _call("fg_softmax_rows_f32_bf16", _ptr(scores), _ptr(probs), scores.shape[0], scores.shape[1], float(scale),
          _stream(scores))
    return probs


def vae_latent_to_cl(z, mean, inv_std):
    """(C,T,H,W) -> (T,H,W,C) de-normalised."""
    _dev(z, "z"), _dev(mean, "mean"), _dev(inv_std, "inv_std")
    assert z.dim() == 4 and z.is_contiguous()
    c, t, h, w = z.shape
    out = torch.empty((t, h, w, c), dtype=z.dtype, device=z.device)
    _call("fg_vae_latent_to_cl_bf16", _ptr(z), _ptr(mean), _ptr(inv_std), _ptr(out), c, t, h, w, _stream(z))
    return out


def vae_unpatchify(x, video, t0, clamp):
    """x (T,H,W,12) -> video[:, t0:t0+T] of (3,F,2H,2W)."""
    _dev(x, "x"), _dev(video, "video")
    t, h, w, c = x.shape
    assert c == 12 and x.is_contiguous() and video.is_contiguous()
    assert video.shape[0] == 3 and video.shape[2] == 2 * h and video.shape[3] == 2 * w
    _call("fg_vae_unpatchify_bf16", _ptr(x), _ptr(video), t, h, w, video.shape[1], t0, int(clamp), _stream(x))
    return video


def vae_tile_accumulate(tile, values, weight, y0, x0, border_h, border_w, bounds):
    _dev(tile, "tile"), _dev(values, "values"), _dev(weight, "weight")
    c, f, th, tw = tile.shape
    _, _, hv, wv = values.shape
    assert tile.is_contiguous() and values.is_contiguous() and weight.is_contiguous() and values.shape[0] == c
    bits = sum(1 << i for i, bnd in enumerate(bounds) if bnd)
    _call("fg_vae_tile_accumulate_bf16", _ptr(tile), _ptr(values), _ptr(weight), c, f, hv, wv, th, tw, y0, x0,
          border_h, border_w, bits, _stream(tile))


def vae_tile_finalize(values, weight, clamp=True):
    c, f, hv, wv = values.shape
    _call("fg_vae_tile_finalize_bf16", _ptr(values), _ptr(weight), c, f, hv, wv, int(clamp), _stream(values))
    return values


def vae_patchify(video):
    """(3,T,H,W) -> (T,H/2,W/2,16) channels-last (12 patch channels + 4 zero channels)."""
    _dev(video, "video")
    assert video.dim() == 4 and video.shape[0] == 3 and video.is_contiguous()
    _, t, h, w = video.shape
    out = torch.empty((t, h // 2, w // 2, 16), dtype=video.dtype, device=video.device)
    _call("fg_vae_patchify_bf16", _ptr(video), _ptr(out), t, h, w, _stream(video))
    return out


def avgdown3d_add(x, main, ft, fs):
    _dev(x, "x"), _dev(main, "main")
    t, h, w, cin = x.shape
    cout = main.shape[-1]
    assert tuple(main.shape) == ((t + ft - 1) // ft, h // fs, w // fs, cout) and x.is_contiguous() and main.is_contiguous()
    out = torch.empty_like(main)
    _call("fg_avgdown3d_add_bf16", _ptr(x), _ptr(main), _ptr(out), t, h, w, cin, cout, ft, fs, _stream(x))
    return out


def vae_latent_from_cl(x, mean, inv_std, z_dim):
    """(T,h,w,Cx) -> (z_dim,T,h,w) normalised latent."""
    _dev(x, "x"), _dev(mean, "mean"), _dev(inv_std, "inv_std")
    t, h, w, cx = x.shape
    assert x.is_contiguous()
    out = torch.empty((z_dim, t, h, w), dtype=x.dtype, device=x.device)
    _call("fg_vae_latent_from_cl_bf16", _ptr(x), _ptr(mean), _ptr(inv_std), _ptr(out), z_dim, cx, t, h, w, _stream(x))
    return out


def video_to_uint8(video):
    """(3,F,H,W) bf16 in [-1,1] -> (F,H,W,3) uint8."""
    _dev(video, "video")
    assert video.is_contiguous()
    _, f, h, w = video.shape
    out = torch.empty((f, h, w, 3), dtype=torch.uint8, device=video.device)
    _call("fg_video_to_uint8", _ptr(video), _ptr(out), f, h, w, _stream(video))
    return out


# ----------------------------------------------------------------------------- umT5 text encoder helpers
def softmax_bias(scores, bias, key_mask=None):
    """scores, bias (rows, cols) bf16; key_mask (cols,) int32 or None -> probs bf16."""
    _dev(scores, "scores"), _dev(bias, "bias")
    assert scores.dim() == 2 and scores.shape == bias.shape and scores.is_contiguous() and bias.is_contiguous()
    if key_mask is not None:
        _dev(key_mask, "key_mask", torch.int32)
        assert key_mask.shape == (scores.shape[1],) and key_mask.is_contiguous()
    probs = torch.empty_like(scores)
    _call("fg_softmax_bias_bf16", _ptr(scores), _ptr(bias), _ptr(key_mask), _ptr(probs), scores.shape[0], scores.shape[1],
          _stream(scores))
    return probs


def gated_gelu(fc1, gate):
    _dev(fc1, "fc1"), _dev(gate, "gate")
    assert fc1.shape == gate.shape and fc1.is_contiguous() and gate.is_contiguous()
    out = torch.empty_like(fc1)
    _call("fg_gated_gelu_bf16", _ptr(fc1), _ptr(gate), _ptr(out), fc1.numel(), _stream(fc1))
    return out
