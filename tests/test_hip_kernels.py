"""GPU parity: every HIP kernel, called through the C ABI (fairygen_amd.hip), against the CPU oracle.

Tolerances (floating point; stated per test):
  * elementwise kernels reproduce the reference's bf16 rounding points, so they must match the oracle's bf16
    result to <= 1 bf16 ulp (reduction order inside a LayerNorm/RMSNorm row may flip a rounding);
  * MFMA kernels (attention, conv): error vs the oracle's fp32 result <= 2x the error of the oracle's own
    bf16 result vs fp32 on the same inputs, plus a small floor (SURVEY.md §8d "parity tolerance").
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import seeded
from oracle import wan_dit, wan_vae
from oracle import pipeline as opipe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from fairygen_amd import hip as h
    h.load()
    assert torch.cuda.is_available()
    return h


def dev(t):
    return t.to("cuda")


def assert_gelu_of(got, y, what):
    """got == GELU(tanh) of the bf16 tensor y, rounded to bf16: <= 1 bf16 ulp (floor 1e-3) from torch's, and identical on >= 95 % of the
    elements that are not deep in the negative tail.  (For y < -3 torch's fp32 formula 0.5 y (1 + tanh u) is quantised by the cancellation in
    1 + tanh u — steps of 2^-24 y — while y * rcp(1 + exp2(.)) is not: values of 1e-6 and below that differ in their leading digits.)"""
    want = F.gelu(y.float().cpu().to(torch.bfloat16), approximate="tanh")
    assert_close_bf16(got, want, 1.0, what, max_mismatch=1.0)
    keep = y.float().cpu() > -3.0
    frac = (got.float().cpu()[keep] != want.float()[keep]).float().mean().item()
    assert frac <= 0.05, f"{what}: {frac:.3f} of the elements above the negative tail differ"


def ulp_diff(a, b):
    """max difference in bf16 ulps (as int16 bit patterns of same-sign values) + max abs."""
    a, b = a.float().cpu(), b.float().cpu()
    return (a - b).abs().max().item()


def assert_close_bf16(got, want, rel_ulps=1.0, what="", mag=None, max_mismatch=2e-3):
    """got == want up to `rel_ulps` bf16 ulps of `mag` (default |want|; pass the magnitude of the largest
    rounded intermediate when the result is a cancelling sum: a 1-ulp flip of an intermediate, caused by a
    different fp32 reduction order inside the row statistic, is 1 ulp of THAT magnitude), and bit-identical
    in all but a `max_mismatch` fraction of the elements."""
    got, want = got.float().cpu(), want.float().cpu()
    mag = want.abs() if mag is None else torch.maximum(want.abs(), mag.float().cpu().abs().expand_as(want))
    tol = rel_ulps * (mag.clamp_min(1e-3) * 2.0 ** -7)       # 1 bf16 ulp <= 2^-7 relative
    bad = ((got - want).abs() > tol)
    assert not bad.any(), f"{what}: {bad.sum().item()} / {bad.numel()} beyond {rel_ulps} ulp; max abs {(got - want).abs().max().item()}"
    frac = (got != want).float().mean().item()
    assert frac <= max_mismatch, f"{what}: {frac:.2e} of the elements differ from the oracle (allowed {max_mismatch:.0e})"


# ------------------------------------------------------------------------------------------ DiT elementwise
@pytest.mark.parametrize("rows,C,mod_rows", [(37, 3072, 2), (5, 256, 1), (130, 3072, 130), (64, 1536, 2)])
def test_ln_modulate(hip, rows, C, mod_rows):
    x = seeded((1, rows, C), 1)
    table = seeded((mod_rows, 6, C), 2, scale=0.5)
    first = rows // 3
    mod = hip.ModTable(dev(table), first)
    got = hip.ln_modulate(dev(x), mod, 0, 1, 1e-6)
    idx = torch.zeros(rows, dtype=torch.long) if mod_rows == 1 else (
        (torch.arange(rows) >= first).long() if mod_rows == 2 else torch.arange(rows))
    shift, scale = table[idx, 0].unsqueeze(0), table[idx, 1].unsqueeze(0)
    want = wan_dit.layer_norm(x, 1e-6) * (1 + scale) + shift
    assert_close_bf16(got, want, 1.0, "ln_modulate", mag=shift)


def test_ln_affine_and_gate_residual(hip):
    rows, C = 77, 3072
    x, y = seeded((1, rows, C), 3), seeded((1, rows, C), 4)
    w, b = 1 + 0.1 * seeded((C,), 5), 0.1 * seeded((C,), 6)
    w, b = w.to(torch.bfloat16), b.to(torch.bfloat16)
    assert_close_bf16(hip.ln_affine(dev(x), dev(w), dev(b), 1e-6), wan_dit.layer_norm(x, 1e-6, w, b), 1.0, "ln_affine", mag=b)
    table = seeded((2, 6, C), 7)
    mod = hip.ModTable(dev(table), 30)
    idx = (torch.arange(rows) >= 30).long()
    got = hip.gate_residual(dev(x), dev(y), mod, 2)
    assert torch.equal(got.cpu(), x + table[idx, 2].unsqueeze(0) * y)          # pure elementwise: bit exact
    assert torch.equal(hip.gate_residual(dev(x), dev(y)).cpu(), x + y)


def test_fused_residual_norms(hip):
    rows, C = 45, 3072
    x, y = seeded((1, rows, C), 8), seeded((1, rows, C), 9)
    table, table2 = seeded((2, 6, C), 10, scale=0.5), seeded((2, 6, C), 11, scale=0.5)
    idx = (torch.arange(rows) >= 20).long()
    mod, mod2 = hip.ModTable(dev(table), 20), hip.ModTable(dev(table2), 20)
    x1 = x + table[idx, 5].unsqueeze(0) * y
    xo, no = hip.residual_ln_modulate(dev(x), dev(y), mod, 5, 0, 1, 1e-6, norm_mod=mod2)
    assert torch.equal(xo.cpu(), x1)
    assert_close_bf16(no, wan_dit.layer_norm(x1, 1e-6) * (1 + table2[idx, 1].unsqueeze(0)) + table2[idx, 0].unsqueeze(0), 1.0, "res+modulate",
                      mag=table2[idx, 0].unsqueeze(0))
    xo, no = hip.residual_ln_modulate(dev(x), dev(y), mod, None, 3, 4, 1e-6)
    assert torch.equal(xo.cpu(), x + y)
    w, b = (1 + 0.1 * seeded((C,), 5)).to(torch.bfloat16), (0.1 * seeded((C,), 6)).to(torch.bfloat16)
    xo, no = hip.residual_ln_affine(dev(x), dev(y), dev(w), dev(b), 1e-6, mod, 2)
    x2 = x + table[idx, 2].unsqueeze(0) * y
    assert torch.equal(xo.cpu(), x2)
    assert_close_bf16(no, wan_dit.layer_norm(x2, 1e-6, w, b), 1.0, "res+affine", mag=b)


@pytest.mark.parametrize("M,K,N", [(700, 256, 768),       # 9 tiles: every one a 64-column tail piece (second body, ring of 3)
                                   (4200, 512, 4096),     # 272 tiles: one full round of 256-column tiles + cut tail, ragged last rows
                                   (10500, 128, 2048),    # 336 tiles: left-over tiles NOT cut (10 per XCD), K = one step pair, 4 rows in the last row tile
                                   (3410, 3072, 3072)])   # the 1/8 token shard of the headline clip: 168 tiles on 256 CUs (o / cross-q / cross-o of config 4)
def test_gemm_epilogue(hip, M, K, N):
    """fg_gemm_epilogue_bf16 (persistent MFMA GEMM, nn.Linear of the DiT blocks: models/wan_video_dit.py:130-133,208-209): error to
    the fp32 product <= 2x the error of the reference's own bf16 op (CPU F.linear) + floor; the residual modes (GateModule :188-193,
    :225-228) must equal x + gate * y built from the kernel's own y with the reference's two bf16 roundings, bit for bit."""
    x, w, b = seeded((1, M, K), 41), seeded((N, K), 42, scale=0.05), seeded((N,), 43, scale=0.2)
    ref32 = F.linear(x.float(), w.float(), b.float())
    ref16 = F.linear(x, w, b)
    y = hip.gemm_epilogue(dev(x), dev(w), dev(b))
    assert y.shape == (1, M, N)
    err, ref_err = (y.float().cpu() - ref32).abs().max().item(), (ref16.float() - ref32).abs().max().item()
    assert err <= 2 * ref_err + 1e-3, f"gemm {M}x{K}x{N}: {err} vs reference bf16 error {ref_err}"
    assert (y.cpu() != ref16).float().mean().item() < 0.02          # same rounding almost everywhere (fp32 summation order differs)
    res = seeded((1, M, N), 44)
    first = 130
    table = seeded((2, 6, N), 45)
    idx = (torch.arange(M) >= first).long()
    got = hip.gemm_epilogue(dev(x), dev(w), dev(b), out=dev(res).clone(), residual=True, mod=hip.ModTable(dev(table), first), gate_idx=5)
    assert torch.equal(got.cpu(), res + table[idx, 5].unsqueeze(0) * y.cpu())
    got = hip.gemm_epilogue(dev(x), dev(w), dev(b), out=dev(res).clone(), residual=True, mod=hip.ModTable(dev(table[:1].contiguous())), gate_idx=2)
    assert torch.equal(got.cpu(), res + table[0, 2] * y.cpu())
    got = hip.gemm_epilogue(dev(x), dev(w), dev(b), out=dev(res).clone(), residual=True)
    assert torch.equal(got.cpu(), res + y.cpu())
    # mode 4: nn.Linear then nn.GELU(approximate='tanh') (:208): GELU of the kernel's own bf16 y, rounded again; exp2 / rcp based like
    # fg_act_bf16: <= 1 bf16 ulp from libm's, nearly all elements identical
    got = hip.gemm_epilogue(dev(x), dev(w), dev(b), act="gelu_tanh")
    assert_gelu_of(got, y, "gemm + gelu")
    with pytest.raises(hip.HipLibraryError):
        hip.gemm_epilogue(dev(x), dev(w), dev(b), out=dev(res).clone(), residual=True, act="gelu_tanh")
    with pytest.raises(hip.HipLibraryError):
        hip.gemm_epilogue(dev(x), dev(w), dev(b), out=torch.empty((M, N - 8), dtype=torch.bfloat16, device="cuda"))      # wrong-sized out
    with pytest.raises(hip.HipLibraryError):
        hip.gemm_epilogue(dev(x), dev(w[:576]), dev(b[:576]))                         # N % 256 != 0
    # strided input rows (a column slice of a wider tensor), as the fused QKV output is consumed elsewhere
    wide = seeded((1, M, K + 64), 46)
    y2 = hip.gemm_epilogue(dev(wide)[..., :K], dev(w), dev(b))
    assert torch.equal(y2.cpu(), hip.gemm_epilogue(dev(wide[..., :K].contiguous()), dev(w), dev(b)).cpu())
    with pytest.raises(hip.HipLibraryError):
        hip.gemm_epilogue(dev(x), dev(w)[:, : K - 64].contiguous(), dev(b))          # K mismatch
    with pytest.raises(hip.HipLibraryError):
        hip.gemm_epilogue(x, w, b)                                                   # CPU tensors: no fallback


@pytest.mark.parametrize("M,K,N", [(4200, 6144, 4096),     # 272 tiles: one full round + 2 left-over tiles per XCD, k-split 16 ways; 104 rows in the last row tile
                                   (600, 14336, 3072),     # ffn.2's reduction length: 36 tiles = 4-5 per XCD, no full round, cut 8 / 6 ways
                                   (8300, 14336, 512),     # 66 tiles: 8-9 per XCD on 32 CUs, cut 4 / 3 ways (224 k-steps -> 75 + 75 + 74)
                                   (66200, 6144, 256),     # 259 tiles in one column: one left-over tile (32 pieces) on three XCDs, none on the others
                                   (3410, 14336, 3072)])   # ffn.2 at the 1/8 token shard (config 4): 168 tiles, 21 per XCD — no piece fits: every tile whole
def test_gemm_epilogue_ksplit(hip, M, K, N):
    """The k-split path of fg_gemm_epilogue_bf16 (K >= 6 144 with a workspace: the left-over tiles of an XCD's last round are computed as
    k-range pieces whose fp32 accumulators go to the workspace and are added in k order by gemm_reduce_kernel, csrc/dit_gemm.hip) — the
    default path of ffn.2 (models/wan_video_dit.py:208-209,225-228).  Checked (1) against the fp32 product on sampled rows with the 2x
    criterion, (2) against the same kernel without a workspace (every element one k-ordered accumulation): <= 1 bf16 ulp apart, (3) the
    residual modes: x + gate * y with the reference's two roundings from the kernel's OWN k-split y, bit for bit (the reduce kernel
    re-implements the accumulator lane map, the gate row select and the epilogue)."""
    x, w, b = seeded((1, M, K), 141, scale=0.5), seeded((N, K), 142, scale=0.02), seeded((N,), 143, scale=0.2)
    dx, dw, db = dev(x), dev(w), dev(b)
    y = hip.gemm_epilogue(dx, dw, db)
    y_nows = hip.gemm_epilogue(dx, dw, db, workspace=False)
    rows = torch.cat([torch.arange(0, 300), torch.arange(M // 2, M // 2 + 100), torch.arange(M - 300, M)])
    ref32 = F.linear(x[0, rows].float(), w.float(), b.float())
    ref16 = F.linear(x[0, rows], w, b).float()
    ref_err = (ref16 - ref32).abs().max().item()
    for name, t in (("k-split", y), ("no workspace", y_nows)):
        err = (t[0].cpu()[rows].float() - ref32).abs().max().item()
        assert err <= 2 * ref_err + 1e-3, f"gemm {M}x{K}x{N} ({name}): {err} vs reference bf16 error {ref_err}"
    assert_close_bf16(y, y_nows, 1.0, "k-split vs single accumulation", max_mismatch=0.02)
    if M != 3410:      # (at 3 410 rows the plan has no k-range pieces: the two calls are the same computation)
        assert not torch.equal(y, y_nows), "the k-split path did not run (identical to the single-accumulation result)"
    res = seeded((1, M, N), 144)
    first = M - 300                                            # the gate class changes inside the last rows (left-over tiles live there)
    table = seeded((2, 6, N), 145)
    idx = (torch.arange(M) >= first).long()
    got = hip.gemm_epilogue(dx, dw, db, out=dev(res).clone(), residual=True, mod=hip.ModTable(dev(table), first), gate_idx=5)
    assert torch.equal(got.cpu(), res + table[idx, 5].unsqueeze(0) * y.cpu())
    got = hip.gemm_epilogue(dx, dw, db, out=dev(res).clone(), residual=True, mod=hip.ModTable(dev(table), 130), gate_idx=2)
    assert torch.equal(got.cpu(), res + table[(torch.arange(M) >= 130).long(), 2].unsqueeze(0) * y.cpu())
    got = hip.gemm_epilogue(dx, dw, db, out=dev(res).clone(), residual=True)
    assert torch.equal(got.cpu(), res + y.cpu())
    # strided A rows through the k-split path
    wide = seeded((1, 600, K + 128), 146, scale=0.5)
    dwide = dev(wide)
    assert torch.equal(hip.gemm_epilogue(dwide[..., 64:64 + K], dw, db), hip.gemm_epilogue(dwide[..., 64:64 + K].contiguous(), dw, db))


@pytest.mark.parametrize("M,K,N", [(4200, 512, 4096), (600, 14336, 3072), (700, 256, 768), (27280, 256, 3072)])
def test_gemm_unit_scheduler(hip, M, K, N):
    """The units of a launch (whole tiles, k-range pieces, 64-column pieces) are fixed by the shape; which workgroup computes which is
    decided by per-XCD cursors at run time.  With fewer workgroups than CUs (fg_gemm_debug_grid: what a CU held by another stream's kernel
    looks like) every unit is still computed, by the workgroups that are there: results bit-identical to the full grid's, launch after
    launch (the last workgroup of a launch clears the cursors), in both operand types and with the fused epilogues."""
    lib = hip.load()
    x, w, b = seeded((1, M, K), 161), seeded((N, K), 162, scale=0.05), seeded((N,), 163, scale=0.2)
    dx, dw, db = dev(x), dev(w), dev(b)
    res = dev(seeded((1, M, N), 164))
    mod = hip.ModTable(dev(seeded((2, 6, N), 165)), 130)
    xq, sc = hip.fp8_quant_rows(dx)
    w8 = dw.to(torch.float8_e4m3fn)

    def run():
        return (hip.gemm_epilogue(dx, dw, db), hip.gemm_epilogue(dx, dw, db, out=res.clone(), residual=True, mod=mod, gate_idx=2),
                hip.gemm_epilogue(dx, dw, db, act="gelu_tanh"), hip.gemm_fp8(xq, sc, w8, db) if K % 256 == 0 else None)
    want = run()
    assert torch.isfinite(want[0].float()).all()
    try:
        for grid in (200, 128, 8, 256, 64):
            assert lib.fg_gemm_debug_grid(grid) == 0
            for rep in range(2):
                got = run()
                for i, (g, wnt) in enumerate(zip(got, want)):
                    assert wnt is None or torch.equal(g, wnt), f"grid {grid}, launch {rep}, output {i}"
    finally:
        lib.fg_gemm_debug_grid(0)
    assert all(wnt is None or torch.equal(g, wnt) for g, wnt in zip(run(), want))


def test_gemm_two_streams(hip):
    """Launches on different streams use different cursor blocks (the library keeps one per (device, stream)) and do not wait on each other:
    two streams issuing the persistent GEMM concurrently — its workgroups take CUs as they come free, nothing spins — give the single-stream
    results bit for bit."""
    shapes = [(4200, 512, 4096), (2700, 3072, 3072), (600, 14336, 3072)]
    data = [(dev(seeded((1, m, k), 171 + i)), dev(seeded((n, k), 181 + i, scale=0.05)), dev(seeded((n,), 191 + i, scale=0.2))) for i, (m, k, n) in enumerate(shapes)]
    want = [hip.gemm_epilogue(x, w, b) for x, w, b in data]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [[], []]
    for rep in range(6):
        for si, st in enumerate(streams):
            with torch.cuda.stream(st):
                x, w, b = data[(rep + si) % len(data)]
                got[si].append(((rep + si) % len(data), hip.gemm_epilogue(x, w, b)))
    torch.cuda.synchronize()
    for si in range(2):
        for idx, y in got[si]:
            assert torch.equal(y, want[idx]), f"stream {si}, shape {shapes[idx]}"


@pytest.mark.parametrize("M,K,N", [(700, 3072, 768),        # 9 tiles: 64-column pieces (second body); 24 k-steps of 128 e4m3
                                   (4200, 1024, 4096),      # 272 tiles: a full round + cut tail, ragged last rows, 8 k-steps
                                   (600, 14336, 3072)])     # ffn.2's reduction: 112 k-steps, left-over tiles as k-range pieces + reduce kernel
def test_gemm_fp8(hip, M, K, N):
    """fg_gemm_fp8_bf16 = the torch._scaled_mm call of AutoWrappedLinear.fp8_linear (core/vram/layers.py:343-357) on e4m3 MFMA:
    (a) against the oracle's written-out op (oracle.wan_dit.scaled_mm; pinned by the reference-run 1x1 vectors) on the same fp8 operands:
    e4m3 x e4m3 products are exact in fp32, so only the fp32 summation order differs: <= 1 bf16 ulp, nearly all elements identical;
    (b) against the library op on the device, same criterion; (c) residual / GELU epilogues from the kernel's own y, bit for bit."""
    x = seeded((1, M, K), 151, scale=2.0)
    x[0, 5] *= 300.0                                          # rows above the fp8 range: scale_a > 1
    x[0, M - 3] *= 1000.0
    w, b = seeded((N, K), 152, scale=0.05), seeded((N,), 153, scale=0.3)
    xq, sc = hip.fp8_quant_rows(dev(x))
    assert sc[5].item() > 1.0 and sc[0].item() == 1.0
    w8 = dev(w).to(torch.float8_e4m3fn)
    y = hip.gemm_fp8(xq, sc, w8, dev(b))
    assert y.shape == (M, N) and y.dtype == torch.bfloat16
    rows = torch.cat([torch.arange(0, 160), torch.arange(M - 160, M)])
    want = wan_dit.scaled_mm(xq[rows].cpu(), w8.cpu().T, sc[rows].cpu(), torch.ones((1, N)), b, torch.bfloat16)
    # an output that is a cancelling sum carries the accumulation error of its terms whatever its own magnitude, and the e4m3 MFMA's 64-wide
    # dot product is not an fp32 fma chain: measured (tools/diag/fp8_gemm2.py) acc = -0.3195 exactly and in a CPU fp32 matmul, -0.3193 in this
    # kernel AND in the library's (bit-identical to each other), i.e. ~2^-18 of sum_k |x_q w_q| = 82.  Tolerance = 1 bf16 ulp of
    # max(|y|, 2^-10 * scale_a * sum_k |x_q w_q|), i.e. 2^-17 of the absolute sum
    mag = (xq.float().abs() @ w8.float().abs().T) * sc * 2.0 ** -10
    assert_close_bf16(y[rows], want, 1.0, "fp8 gemm vs oracle", mag=mag[rows], max_mismatch=0.02)
    lib = torch._scaled_mm(xq, w8.T, scale_a=sc, scale_b=torch.ones((1, N), device="cuda"), bias=dev(b), out_dtype=torch.bfloat16)
    assert_close_bf16(y, lib, 1.0, "fp8 gemm vs torch._scaled_mm", mag=mag, max_mismatch=0.02)
    if K >= 12288:
        y_nows = hip.gemm_fp8(xq, sc, w8, dev(b), workspace=False)
        # (usually bit-identical: the pieces end on MFMA block boundaries and the fp32 sums of e4m3 products are mostly exact)
        assert_close_bf16(y, y_nows, 1.0, "k-split vs single accumulation", mag=mag, max_mismatch=0.02)
    res = seeded((M, N), 154)
    first = M - 200
    table = seeded((2, 6, N), 155)
    idx = (torch.arange(M) >= first).long()
    got = hip.gemm_fp8(xq, sc, w8, dev(b), out=dev(res).clone(), residual=True, mod=hip.ModTable(dev(table), first), gate_idx=5)
    assert torch.equal(got.cpu(), res + table[idx, 5] * y.cpu())
    got = hip.gemm_fp8(xq, sc, w8, dev(b), out=dev(res).clone(), residual=True)
    assert torch.equal(got.cpu(), res + y.cpu())
    got = hip.gemm_fp8(xq, sc, w8, dev(b), act="gelu_tanh", lead_shape=(1, M))
    assert got.shape == (1, M, N)
    assert_gelu_of(got[0], y, "fp8 gemm + gelu")
    with pytest.raises(hip.HipLibraryError):
        hip.gemm_fp8(xq, sc[:-1], w8, dev(b))
    with pytest.raises(hip.HipLibraryError):
        hip.gemm_fp8(xq.cpu(), sc.cpu(), w8.cpu(), b)          # CPU tensors: no fallback


@pytest.mark.parametrize("heads,C,grid", [(24, 3072, (2, 3, 5)), (2, 256, (3, 4, 4))])
def test_rmsnorm_rope(hip, heads, C, grid):
    f, h, w_ = grid
    n = f * h * w_
    wide = seeded((1, n, 3 * C), 12)                     # q slice of a fused QKV buffer (ld = 3C)
    wt = (1 + 0.1 * seeded((C,), 13)).to(torch.bfloat16)
    table = wan_dit.rope_table_3d(C // heads, f, h, w_)
    cos, sin = table.real.reshape(n, -1).contiguous(), table.imag.reshape(n, -1).contiguous()
    xq = wide[..., C:2 * C]
    want = wan_dit.rope_apply(wan_dit.rms_norm(xq, wt, 1e-6), table, heads)
    got = hip.rmsnorm_rope(dev(wide)[..., C:2 * C], dev(wt), heads, 1e-6, dev(cos), dev(sin))
    assert_close_bf16(got, want, 1.0, "rmsnorm+rope")
    # the pipeline's default: ONE fp32 interleaved table, rotation as fp32 FMAs — same criterion (<= 1 bf16 ulp, >= 99.8 %
    # of the elements bit-identical to the oracle's fp64 rotation), and next to the fp64 mode of the same kernel
    cs = torch.stack([table.real.reshape(n, -1), table.imag.reshape(n, -1)], dim=-1).to(torch.float32).contiguous()
    got32 = hip.rmsnorm_rope(dev(wide)[..., C:2 * C], dev(wt), heads, 1e-6, dev(cs))
    assert_close_bf16(got32, want, 1.0, "rmsnorm+rope (fp32 table)")
    same = (got32 == got).float().mean().item()
    assert same >= 0.998, f"fp32-table rotation: only {same:.5f} of the elements equal the fp64 mode"
    got = hip.rmsnorm_rope(dev(wide)[..., :C], dev(wt), heads, 1e-6)
    assert_close_bf16(got, wan_dit.rms_norm(wide[..., :C], wt, 1e-6), 1.0, "rmsnorm")


def test_ulysses_packing_kernels(hip):
    """fg_rmsnorm_rope_grouped_bf16 / fg_copy_groups_bf16: the head-group-major layouts around the Ulysses all-to-alls
    are exact re-layouts of the plain kernels' results (bit-exact), including a short (padded) token shard."""
    heads, C, P, n, size = 4, 512, 2, 15, 17            # 2 head groups of 256 columns; 15 tokens in a 17-row chunk
    g = C // P
    f, h, w_ = 1, 3, 5
    qkv = dev(seeded((1, n, 3 * C), 31))
    wt = dev((1 + 0.1 * seeded((C,), 32)).to(torch.bfloat16))
    table = wan_dit.rope_table_3d(C // heads, f, h, w_)
    cos, sin = dev(table.real.reshape(n, -1).contiguous()), dev(table.imag.reshape(n, -1).contiguous())
    send = torch.full((P, size, 3, g), 7.0, dtype=torch.bfloat16, device="cuda")
    flat, layout = send.view(-1), (g, size * 3 * g, 3 * g)
    hip.rmsnorm_rope(qkv[..., :C], wt, heads, 1e-6, cos, sin, grouped=(flat, *layout))
    hip.rmsnorm_rope(qkv[..., C:2 * C], wt, heads, 1e-6, grouped=(flat[g:], *layout))
    hip.copy_groups(qkv.view(-1)[2 * C:], g, 3 * C, flat[2 * g:], size * 3 * g, 3 * g, P, n, g)
    q = hip.rmsnorm_rope(qkv[..., :C], wt, heads, 1e-6, cos, sin)[0]
    k = hip.rmsnorm_rope(qkv[..., C:2 * C], wt, heads, 1e-6)[0]
    for j, t in enumerate((q, k, qkv[0, :, 2 * C:])):
        assert torch.equal(send[:, :n, j], t.unflatten(-1, (P, g)).transpose(0, 1)), f"send block {j}"
    cs = torch.stack([cos, sin], dim=-1).to(torch.float32).contiguous()      # the fp32-table mode, grouped == plain
    send32 = torch.full((P, size, 3, g), 7.0, dtype=torch.bfloat16, device="cuda")
    hip.rmsnorm_rope(qkv[..., :C], wt, heads, 1e-6, cs, grouped=(send32.view(-1), *layout))
    q32 = hip.rmsnorm_rope(qkv[..., :C], wt, heads, 1e-6, cs)[0]
    assert torch.equal(send32[:, :n, 0], q32.unflatten(-1, (P, g)).transpose(0, 1))
    assert (send[:, n:] == 7.0).all(), "rows past the shard must not be written"
    blocks = dev(seeded((P, size, g), 33))               # received head-group blocks -> "b s (n d)" rows
    a = torch.empty((1, n, C), dtype=torch.bfloat16, device="cuda")
    hip.copy_groups(blocks.view(-1), size * g, g, a.view(-1), g, C, P, n, g)
    assert torch.equal(a[0], blocks[:, :n].transpose(0, 1).reshape(n, C))
    with pytest.raises(hip.HipLibraryError):
        hip.copy_groups(blocks.view(-1), size * g, g, a.view(-1)[:-8], g, C, P, n, g)


@pytest.mark.parametrize("C,act", [(3072, None), (14336, "gelu_tanh"), (256, None), (512, "gelu_tanh")])
def test_fp8_quant_rows(hip, C, act):
    """fg_fp8_quant_rows_bf16 == the reference's fp8_linear activation path (core/vram/layers.py:331-342), bit for bit:
    scale_a = clamp(bf16(max|x| / 448), 1), x / (scale_a + 1e-8) in fp32, RNE cast to torch.float8_e4m3fn."""
    rows = 37
    wide = seeded((1, rows, C + 64), 111, scale=2.0)
    wide[0, 5] *= 400.0                                   # rows above the fp8 range get a scale > 1
    wide[0, 9] *= 1e-3                                    # and tiny rows exercise the subnormal roundings
    x = wide[..., 32:32 + C]                              # strided slice (ld = C + 64)
    xa = F.gelu(x, approximate="tanh") if act else x
    x2 = xa.reshape(-1, C)
    want_scale = torch.clamp(x2.abs().amax(-1, keepdim=True) / 448.0, min=1.0).float()
    want = (x2 / (want_scale + 1e-8)).to(torch.float8_e4m3fn)
    got, scale = hip.fp8_quant_rows(dev(wide)[..., 32:32 + C], act)
    assert got.dtype == torch.float8_e4m3fn and got.shape == (rows, C) and scale.shape == (rows, 1)
    assert torch.equal(scale.cpu(), want_scale)
    assert want_scale[5].item() > 1.0 and want_scale[0].item() == 1.0
    same = got.cpu().view(torch.uint8) == want.view(torch.uint8)
    if act is None:
        assert same.all(), f"{(~same).sum().item()} fp8 bytes differ"
    else:       # __expf-based GELU vs libm: rare 1-ulp bf16 flips before the quantisation (same allowance as fg_act_bf16)
        assert same.float().mean().item() > 0.95
        assert (got.cpu().float() - want.float()).abs().max().item() <= 0.07 * want.float().abs().max().item()


def test_fp8_output_norms(hip):
    """fg_ln_modulate_fp8_bf16 / fg_residual_ln_fp8_bf16: the norm kernels of the fp8 Linear mode hand the normalised row over as
    (e4m3 rows, row scales); must equal fg_fp8_quant_rows_bf16 applied to the bf16 result of the plain norm kernels, byte for byte
    (fp8_linear's activation side, core/vram/layers.py:331-342, after models/wan_video_dit.py:224-228)."""
    rows, C = 77, 3072
    x, y = seeded((1, rows, C), 51), seeded((1, rows, C), 52)
    table = seeded((2, 6, C), 53, scale=0.5)
    table[1, 0, 7] = table[0, 3, 9] = 900.0                     # shifts that push row maxima beyond fp8_max: scales > 1 on those rows
    mod = hip.ModTable(dev(table), 30)
    w, b = (1 + 0.1 * seeded((C,), 54)).to(torch.bfloat16), (0.1 * seeded((C,), 55)).to(torch.bfloat16)
    b[11] = 900.0

    def same(pair, bf16_rows):
        q, sc = hip.fp8_quant_rows(bf16_rows)
        assert torch.equal(pair[0].view(torch.uint8).cpu(), q.view(torch.uint8).cpu()) and torch.equal(pair[1].cpu(), sc.cpu())
        assert pair[1].max().item() > 1.0

    same(hip.ln_modulate_fp8(dev(x), mod, 0, 1, 1e-6), hip.ln_modulate(dev(x), mod, 0, 1, 1e-6))
    xo, no = hip.residual_ln_modulate(dev(x), dev(y), mod, 5, 0, 1, 1e-6)
    xo8, pair = hip.residual_ln_modulate_fp8(dev(x), dev(y), mod, 5, 0, 1, 1e-6)
    assert torch.equal(xo8, xo)
    same(pair, no)
    xo, no = hip.residual_ln_affine(dev(x), dev(y), dev(w), dev(b), 1e-6, mod, 2)
    xo8, pair = hip.residual_ln_affine_fp8(dev(x), dev(y), dev(w), dev(b), 1e-6, mod, 2)
    assert torch.equal(xo8, xo)
    same(pair, no)
    xo, no = hip.residual_ln_modulate(dev(x), dev(y), mod, None, 3, 4, 1e-6)
    xo8, pair = hip.residual_ln_modulate_fp8(dev(x), dev(y), mod, None, 3, 4, 1e-6)
    assert torch.equal(xo8, xo)
    same(pair, no)


def test_activations_and_cfg_euler(hip):
    x = seeded((3, 1000, 8), 14, scale=3.0)
    assert_close_bf16(hip.activation(dev(x).clone(), "silu"), F.silu(x), 1.0, "silu", max_mismatch=0.05)          # __expf vs libm: rare 1-ulp flips
    assert_close_bf16(hip.activation(dev(x).clone(), "gelu_tanh"), F.gelu(x, approximate="tanh"), 1.0, "gelu", max_mismatch=0.05)
    lat, p, n_ = seeded((1, 48, 3, 8, 8), 15), seeded((1, 48, 3, 8, 8), 16), seeded((1, 48, 3, 8, 8), 17)
    sig, _ = opipe.wan_sigmas(4)
    for i in range(4):
        pred = n_ + 5.0 * (p - n_)
        want = opipe.euler_step(pred, i, lat, sig)
        ds = float((0 if i == 3 else sig[i + 1]) - sig[i])
        assert torch.equal(hip.cfg_euler(dev(lat), dev(p), dev(n_), 5.0, ds).cpu(), want), f"cfg+euler step {i}"
    want = opipe.euler_step(p, 1, lat, sig)
    assert torch.equal(hip.cfg_euler(dev(lat), dev(p), None, 1.0, float(sig[2] - sig[1])).cpu(), want)


# ------------------------------------------------------------------------------------------------ attention
def _attn_case(hip, nq, nkv, heads, seed, scale_q=1.0):
    c = heads * 128
    q, k, v = seeded((1, nq, c), seed, scale=scale_q), seeded((1, nkv, c), seed + 1), seeded((1, nkv, c), seed + 2)
    ref32 = wan_dit.attention(q.float(), k.float(), v.float(), heads)
    ref16 = wan_dit.attention(q, k, v, heads).float()
    got = hip.attention(dev(q), dev(k), dev(v), heads).float().cpu()
    err_ref = (ref16 - ref32).abs().max().item()
    err = (got - ref32).abs().max().item()
    assert err <= 2 * err_ref + 2e-3, f"attention nq={nq} nkv={nkv}: err {err} vs reference-bf16 err {err_ref}"
    return err, err_ref


@pytest.mark.parametrize("nq,nkv,heads", [(256, 64, 1), (300, 77, 2), (513, 512, 3), (64, 1000, 2), (1560, 1560, 2), (31, 5, 1)])
def test_attention_shapes(hip, nq, nkv, heads):
    _attn_case(hip, nq, nkv, heads, 20)


@pytest.mark.parametrize("nq,nkv,heads,mode", [(300, 1000, 24, "all"), (13618, 2000, 24, "tail"), (3410 // 4, 2700, 24, "all")])
def test_attention_split_kv(hip, nq, nkv, heads, mode):
    """Shapes for which the launcher cuts q-blocks into KV ranges merged by the combine kernel (work balancing)."""
    import ctypes
    R, S = ctypes.c_int(), ctypes.c_int()
    lib = hip.load()
    assert lib.fg_attn_split_choice(1, nq, nkv, heads, lib.fg_attn_workspace_bytes(1, nq, nkv, heads), ctypes.byref(R), ctypes.byref(S)) == 0
    nqb = (nq + 255) // 256
    assert S.value > 1 and R.value == (nqb if mode == "all" else 1), (R.value, S.value)
    _attn_case(hip, nq, nkv, heads, 25)


def test_attention_peaked_softmax(hip):
    # large-magnitude queries: near one-hot softmax rows exercise the running-max rescale path
    _attn_case(hip, 200, 333, 2, 30, scale_q=8.0)


def _spiked_keys(q, k, nkv, picks):
    """Key rows aligned with a few query rows at growing magnitude, late in the sequence: the running max of those rows (every head)
    jumps by ~15-40 log2 units at those tiles, far beyond the kernels' deferred-rescale threshold of 2^6 (tools/attn_ab.py --spike)."""
    k = k.clone()
    for i, (pos, row) in enumerate(picks):
        k[0, pos] = (q[0, row].float() * (1.5 + i)).to(torch.bfloat16)
    return k


@pytest.mark.parametrize("form", ["plain", "pow2"])
@pytest.mark.parametrize("nq,nkv,heads,scale_q", [(300, 1500, 2, 1.0),        # one q-block + a ragged one; 1500 = 23 tiles + 28 keys
                                                   (300, 1500, 2, 8.0),        # peaked rows AND spikes
                                                   (700, 2700, 24, 8.0)])      # every q-block cut into KV ranges (split-KV + merge)
def test_attention_w4_deferred_rescale(hip, nq, nkv, heads, scale_q, form):
    """attn_fwd_w4_kernel (the self-attention kernel: Nkv > 1024, csrc/attention.hip dispatch) on inputs whose running max moves by
    more than 2^6 between tiles, so its out-of-line rescale block (O, l and the pending score tile rescaled once) executes — in the
    first tiles, mid-sequence, in the last full tile and in the ragged last tile.  flash_attention of models/wan_video_dit.py:54-59;
    criterion: error to the oracle's fp32 result <= 2x the error of the oracle's own bf16 result + floor.
    Both bodies of the kernel: "plain" (scale = 1/sqrt(d): exponent argument by v_fma on the fp32 scores) and "pow2" (the scale the
    pipeline passes for self-attention, scale * log2(e) = 2^-3: Q^T pre-multiplied exactly, -max as the MFMA's C operand)."""
    c = heads * 128
    scale, fold = (None, 1.0) if form == "plain" else hip.pow2_softmax_scale(128)
    assert form == "plain" or abs(scale * 1.4426950408889634 - 0.125) < 1e-9
    q, k, v = seeded((1, nq, c), 130, scale=scale_q), seeded((1, nkv, c), 131), seeded((1, nkv, c), 132)
    last_full = (nkv // 64) * 64 - 3
    k = _spiked_keys(q, k, nkv, [(70, 5), (nkv // 3, 17), (nkv // 2, 150), (last_full, 255), (nkv - 2, 299), (nkv - 1, 5)])
    if heads == 24:
        import ctypes
        R, S = ctypes.c_int(), ctypes.c_int()
        lib = hip.load()
        assert lib.fg_attn_split_choice(1, nq, nkv, heads, lib.fg_attn_workspace_bytes(1, nq, nkv, heads), ctypes.byref(R), ctypes.byref(S)) == 0
        assert S.value > 1, "this shape is meant to take the split-KV path"
    # softmax(scale q k^T) = softmax((q / fold) k^T / sqrt(d)): the fp32 reference of the pow2 form is the oracle on q / fold (fp32);
    # the yardstick is the error of the oracle's own bf16 op on the same (equally peaked) problem
    ref32 = wan_dit.attention(q.float() / fold, k.float(), v.float(), heads)
    ref16 = wan_dit.attention(q, k, v, heads).float()
    got = hip.attention(dev(q), dev(k), dev(v), heads, scale=scale).float().cpu()
    assert torch.isfinite(got).all()
    err_ref = (ref16 - wan_dit.attention(q.float(), k.float(), v.float(), heads)).abs().max().item()
    err = (got - ref32).abs().max().item()
    assert err <= 2 * err_ref + 2e-3, f"w4 attention ({form}), spiked keys: err {err} vs reference-bf16 err {err_ref}"
    # the spiked rows themselves are near one-hot on their key: check them separately so that a wrong rescale cannot hide in a max
    for row in (5, 17, 150, 255, 299):
        e = (got[0, row] - ref32[0, row]).abs().max().item()
        assert e <= 2 * err_ref + 2e-3, f"row {row}: {e}"


def test_attention_w4_strided_qkv(hip):
    """The production layout of self-attention: q, k, v are the three column slices of ONE (N, 3C) GEMM output (ld = 3C), Nkv > 1024 so
    the w4 kernel runs; must equal the call on contiguous copies bit for bit, and satisfy the oracle criterion."""
    heads, n = 2, 1337
    c = heads * 128
    qkv = seeded((1, n, 3 * c), 140)
    d = dev(qkv)
    got = hip.attention(d[..., :c], d[..., c:2 * c], d[..., 2 * c:], heads)
    same = hip.attention(d[..., :c].contiguous(), d[..., c:2 * c].contiguous(), d[..., 2 * c:].contiguous(), heads)
    assert torch.equal(got, same)
    q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
    ref32 = wan_dit.attention(q.float(), k.float(), v.float(), heads)
    err_ref = (wan_dit.attention(q, k, v, heads).float() - ref32).abs().max().item()
    assert (got.float().cpu() - ref32).abs().max().item() <= 2 * err_ref + 2e-3


def test_attention_strided_inputs(hip):
    heads, n = 2, 150
    c = heads * 128
    qkv = seeded((1, n, 3 * c), 40)
    q = qkv[..., :c].contiguous()
    want = wan_dit.attention(q.float(), qkv[..., c:2 * c].float(), qkv[..., 2 * c:].float(), heads)
    d = dev(qkv)
    got = hip.attention(dev(q), d[..., c:2 * c], d[..., 2 * c:], heads).float().cpu()
    assert (got - want).abs().max().item() < 2e-2


def test_attention_rejects_cpu_and_bad_head_dim(hip):
    q = seeded((1, 8, 256), 1)
    with pytest.raises(hip.HipLibraryError):
        hip.attention(q, q, q, 2)                       # CPU tensors: no fallback
    with pytest.raises(hip.HipLibraryError):
        hip.attention(dev(q), dev(q), dev(q), 4)        # head_dim 64 unsupported


# ------------------------------------------------------------------------------------------------ VAE kernels
def _cl(x):      # (1,C,T,H,W) -> (T,H,W,C)
    return x[0].permute(1, 2, 3, 0).contiguous()


def _ncthw(x):   # (T,H,W,C) -> (1,C,T,H,W)
    return x.permute(3, 0, 1, 2).unsqueeze(0).contiguous()


@pytest.mark.parametrize("cin,cout,kt,ks,T,H,W,cache", [
    (48, 64, 3, 3, 1, 6, 10, False), (64, 128, 3, 3, 2, 9, 7, True), (128, 256, 3, 1, 1, 5, 6, True),
    (96, 48, 1, 1, 3, 4, 4, False), (256, 12, 3, 3, 4, 8, 8, True), (1024, 1024, 3, 3, 1, 4, 6, True)])
def test_conv3d_cl(hip, cin, cout, kt, ks, T, H, W, cache):
    x = seeded((1, cin, T, H, W), 50)
    w = seeded((cout, cin, kt, ks, ks), 51, scale=(cin * kt * ks * ks) ** -0.5)
    b = seeded((cout,), 52, scale=0.1)
    prev = seeded((1, cin, 2, H, W), 53) if cache else None
    sd = {"c.weight": w, "c.bias": b}
    ref16 = wan_vae.causal_conv3d(sd, "c", x, prev).float()
    sd32 = {k: v.float() for k, v in sd.items()}
    ref32 = wan_vae.causal_conv3d(sd32, "c", x.float(), None if prev is None else prev.float())
    packed = hip.conv_pack_weight(dev(w))
    xin = x if kt == 1 else torch.cat([prev if prev is not None else torch.zeros_like(x[:, :, :1]).expand(-1, -1, 2, -1, -1), x], dim=2)
    got = hip.conv3d_cl(dev(_cl(xin)), packed, dev(b), cout, kt, ks)
    got = _ncthw(got.cpu()).float()
    err_ref, err = (ref16 - ref32).abs().max().item(), (got - ref32).abs().max().item()
    assert err <= 2 * err_ref + 1e-3, f"conv err {err} vs reference-bf16 err {err_ref}"


def test_conv_upsample_interleave_residual(hip):
    c = 64
    x = seeded((1, c, 2, 5, 6), 60)
    w2 = seeded((c, c, 3, 3), 61, scale=(c * 9) ** -0.5)
    b2 = seeded((c,), 62, scale=0.1)
    res = seeded((1, c, 2, 10, 12), 63)
    y = F.interpolate(x[0].permute(1, 0, 2, 3).float(), scale_factor=(2.0, 2.0), mode="nearest-exact").to(torch.bfloat16)
    want = F.conv2d(y, w2, b2, padding=1).permute(1, 0, 2, 3).unsqueeze(0) + res
    got = hip.conv3d_cl(dev(_cl(x)), hip.conv_pack_weight(dev(w2)), dev(b2), c, 1, 3, upsample2x=True, residual=dev(_cl(res)))
    assert (_ncthw(got.cpu()).float() - want.float()).abs().max().item() < 3e-2
    # time_conv + channel->time interleave (Resample.forward :147-156)
    wt = seeded((2 * c, c, 3, 1, 1), 64, scale=(c * 3) ** -0.5)
    bt = seeded((2 * c,), 65, scale=0.1)
    prev = seeded((1, c, 2, 5, 6), 66)
    t = wan_vae.causal_conv3d({"c.weight": wt, "c.bias": bt}, "c", x, prev)
    t = t.reshape(1, 2, c, 2, 5, 6)
    want = torch.stack((t[:, 0], t[:, 1]), 3).reshape(1, c, 4, 5, 6)
    got = hip.conv3d_cl(dev(_cl(torch.cat([prev, x], dim=2))), hip.conv_pack_weight(dev(wt)), dev(bt), 2 * c, 3, 1, time_interleave=True)
    assert (_ncthw(got.cpu()).float() - want.float()).abs().max().item() < 3e-2


@pytest.mark.parametrize("C,silu", [(1024, True), (256, True), (32, False), (512, True)])
def test_vae_rmsnorm_silu(hip, C, silu):
    x = seeded((1, C, 2, 5, 7), 70, scale=2.0)
    g = (1 + 0.1 * seeded((C, 1, 1, 1), 71)).to(torch.bfloat16)
    want = wan_vae.rms_norm_c({"n.gamma": g}, "n", x)
    want = F.silu(want) if silu else want
    got = hip.vae_rmsnorm_silu(dev(_cl(x)), dev(g.view(-1)), silu)
    assert_close_bf16(_ncthw(got.cpu()), want, 1.0, "vae rmsnorm")


@pytest.mark.parametrize("cin,cout,ft,fs,first", [(64, 64, 2, 2, True), (64, 64, 2, 2, False), (64, 32, 1, 2, False), (128, 64, 1, 2, True)])
def test_dupup3d_add(hip, cin, cout, ft, fs, first):
    x = seeded((1, cin, 2, 3, 4), 80)
    sc = wan_vae.dup_up3d(x, cout, ft, fs, first)
    main = seeded(tuple(sc.shape), 81)
    got = hip.dupup3d_add(dev(_cl(x)), dev(_cl(main)), cout, ft, fs, first)
    assert torch.equal(_ncthw(got.cpu()), main + sc)


def test_softmax_latent_unpatchify_uint8(hip):
    s = seeded((37, 301), 90, torch.float32, scale=20.0)
    want = torch.softmax(s * 0.25, dim=-1)
    assert (hip.softmax_rows(dev(s), 0.25).float().cpu() - want).abs().max().item() < 4e-3
    z = seeded((1, 48, 2, 3, 5), 91)
    mean, inv_std = torch.tensor(wan_vae.VAE38_MEAN).to(torch.bfloat16), (1.0 / torch.tensor(wan_vae.VAE38_STD)).to(torch.bfloat16)
    want = z / inv_std.view(1, 48, 1, 1, 1) + mean.view(1, 48, 1, 1, 1)
    assert torch.equal(_ncthw(hip.vae_latent_to_cl(dev(z[0].contiguous()), dev(mean), dev(inv_std)).cpu()), want)
    x = seeded((1, 12, 3, 4, 5), 92, scale=0.8)
    video = torch.zeros((3, 5, 8, 10), dtype=torch.bfloat16, device="cuda")
    hip.vae_unpatchify(dev(_cl(x)), video, 2, True)
    assert torch.equal(video[:, 2:5].cpu(), wan_vae.unpatchify2(x)[0].clamp(-1, 1))
    vid = seeded((3, 2, 8, 8), 93, scale=0.7).clamp(-1.2, 1.2)
    assert torch.equal(hip.video_to_uint8(dev(vid)).cpu(), opipe.video_to_uint8(vid))


def test_encoder_kernels(hip):
    """stride-2 conv (ZeroPad2d((0,1,0,1)) + Conv2d stride 2), patchify, AvgDown3D + add, latent tail."""
    c = 64
    x = seeded((1, c, 2, 10, 12), 110)
    w2, b2 = seeded((c, c, 3, 3), 111, scale=(c * 9) ** -0.5), seeded((c,), 112, scale=0.1)
    y = x[0].permute(1, 0, 2, 3)
    want = F.conv2d(F.pad(y, (0, 1, 0, 1)), w2, b2, stride=2).permute(1, 0, 2, 3).unsqueeze(0)
    got = hip.conv3d_cl(dev(_cl(x)), hip.conv_pack_weight(dev(w2)), dev(b2), c, 1, 3, downsample2x=True)
    assert (_ncthw(got.cpu()).float() - want.float()).abs().max().item() < 3e-2
    vid = seeded((1, 3, 2, 8, 12), 113, scale=0.5)
    p = hip.vae_patchify(dev(vid[0].contiguous())).cpu()
    assert torch.equal(_ncthw(p[..., :12]), wan_vae.patchify2(vid)) and not p[..., 12:].any()
    for cin, cout, ft, fs, T in ((32, 32, 1, 2, 1), (32, 64, 2, 2, 1), (32, 64, 2, 2, 4), (64, 64, 1, 1, 2)):
        xs = seeded((1, cin, T, 6, 8), 114)
        sc = wan_vae.avg_down3d(xs, cout, ft, fs)
        main = seeded(tuple(sc.shape), 115)
        got = hip.avgdown3d_add(dev(_cl(xs)), dev(_cl(main)), ft, fs)
        assert_close_bf16(_ncthw(got.cpu()), main + sc, 1.0, f"avgdown {cin}->{cout} ft{ft} fs{fs}", mag=main, max_mismatch=0.02)
    mean, inv_std = torch.tensor(wan_vae.VAE38_MEAN).to(torch.bfloat16), (1.0 / torch.tensor(wan_vae.VAE38_STD)).to(torch.bfloat16)
    h96 = seeded((1, 96, 1, 3, 5), 116)
    want = (h96[:, :48] - mean.view(1, 48, 1, 1, 1)) * inv_std.view(1, 48, 1, 1, 1)
    assert torch.equal(hip.vae_latent_from_cl(dev(_cl(h96)), dev(mean), dev(inv_std), 48).cpu().unsqueeze(0), want)


def test_text_encoder_kernels(hip):
    from oracle import wan_text
    heads, L = 3, 37
    scores, bias = seeded((heads * L, L), 120, scale=4.0), seeded((heads * L, L), 121)
    mask = torch.zeros(L, dtype=torch.int32); mask[:29] = 1
    b = bias.view(1, heads, L, L).clone()
    b.masked_fill_(mask.view(1, 1, 1, -1) == 0, torch.finfo(torch.bfloat16).min)
    want = torch.softmax((scores.view(1, heads, L, L) + b).float(), dim=-1).to(torch.bfloat16).view(heads * L, L)
    got = hip.softmax_bias(dev(scores), dev(bias), dev(mask))
    assert_close_bf16(got, want, 1.0, "softmax_bias", max_mismatch=0.02)
    assert not got[:, 29:].any()
    fc1, gate = seeded((5, 64, 8), 122), seeded((5, 64, 8), 123, scale=2.0)
    assert_close_bf16(hip.gated_gelu(dev(fc1), dev(gate)), fc1 * wan_text.gelu_tanh_explicit(gate), 1.0, "gated gelu", max_mismatch=0.02)


def test_tile_blend(hip):
    H, W, up, T = 5, 7, 2, 2
    F_ = 4 * T - 3
    tile_size, tile_stride = (3, 4), (2, 3)
    values = torch.zeros((3, F_, H * up, W * up), dtype=torch.bfloat16)
    weight = torch.zeros((F_, H * up, W * up), dtype=torch.bfloat16)
    dv, dw = dev(values), dev(weight)
    for i, (h, h_, w, w_) in enumerate(wan_vae.tile_tasks(H, W, tile_size, tile_stride)):
        th, tw = (min(h_, H) - h) * up, (min(w_, W) - w) * up
        tile = seeded((3, F_, th, tw), 100 + i)
        bounds = (h == 0, h_ >= H, w == 0, w_ >= W)
        border = ((tile_size[0] - tile_stride[0]) * up, (tile_size[1] - tile_stride[1]) * up)
        m = wan_vae.tile_mask(th, tw, bounds, border).to(torch.bfloat16)[0, 0]
        values[:, :, h * up:h * up + th, w * up:w * up + tw] += tile * m
        weight[:, h * up:h * up + th, w * up:w * up + tw] += m[0]
        hip.vae_tile_accumulate(dev(tile), dv, dw, h * up, w * up, border[0], border[1], bounds)
    assert torch.equal(dv.cpu(), values) and torch.equal(dw.cpu(), weight)
    hip.vae_tile_finalize(dv, dw)
    assert torch.equal(dv.cpu(), (values / weight).clamp_(-1, 1))
