#!/usr/bin/env python3
"""Generator of the hand-scheduled body of gemm_w4_kernel (csrc/dit_gemm.hip): C[M,N] = A[M,K] W[N,K]^T + bias (bf16 in / out,
fp32 accumulate), the DiT Linear layers, in the 4-wave / 512-register form of gen_attn_w4.py (whose emitter and gap scheduler
it reuses).

    python3 gen_gemm_w4.py [--stamp] > gemm_w4_asm.inc

Tile: 256 output columns (W rows: the MFMA A operand) x 256 output rows (A rows: the MFMA B operand) x 64 k per step; 4 waves
as 2 (n) x 2 (m), each 128 x 128 = 4 x 4 accumulators of v_mfma_f32_32x32x16_bf16 = all 256 AGPRs; the accumulator holds, per
lane, ONE output row and 4 consecutive columns per register group (8-byte stores, lane-local bias).  Operand tiles go global
-> LDS by LDS-DMA (1 KiB per wave-instruction, XOR swizzle applied to the source chunk) into 2 stages of 64 KiB; fragments
are read one k-step ahead of their MFMAs into two register sets, and the last k-step of a stage is multiplied AFTER the
barrier beside the first fragment reads of the next stage, so neither LDS latency nor the barrier is exposed; every fragment
read feeds 4 MFMAs.  One vmcnt(0) + one barrier per 64 MFMAs.
"""
import argparse
import sys

from gen_attn_w4 import Emitter, Item, schedule, vr, ar, sr

DMA_EVERY_X2 = 4                    # LDS-DMA pieces are issued every DMA_EVERY_X2 / 2 MFMA gaps from the start of a K-step
STAGE, XOFF = 65536, 32768          # LDS: stage s at s*64 KiB: W tile (32 KiB) then X tile (32 KiB)
SB = 40
S = {k: v + SB for k, v in dict(XD=0, WD=4, CD=8, BD=12, WAVE=16, NK=17, LDA=18, LDC=19, KOFF=20, KB=21, T=22, TMP0=23, TMP1=24,
                                TMP2=25, WDST=26, XDST=27, KW=28, TMP64=30, ST0=32, ST1=34, ST2=36, FLAGS=38).items()}
NSREG = 40
# VGPR map (v0.. owned by the body)
F0, F1 = 0, 32                       # fragment sets: W frag ni at +4ni, X frag mi at +16+4mi
V_WRD, V_XRD = 64, 72                # [stage][ks]: 8 + 8 LDS read bases
V_DW, V_DX = 80, 88                  # 8 + 8 LDS-DMA source offsets (pieces of this wave)
V_T = 96                             # temporaries 96..127


def acc(ni, mi, j=0):
    return (ni * 4 + mi) * 16 + j


def emit_inputs(E):
    """%0 x tile base, %1 w tile base, %2 c tile base, %3 bias base (64-bit); %4 x bytes, %5 c bytes, %6 wave, %7 K/64,
    %8 lda bytes, %9 ldc bytes, %10 K bytes (row stride of W), %11 flags."""
    for n, base in enumerate((S["XD"], S["WD"], S["CD"], S["BD"])):
        E.e(f"s_mov_b64 {sr(base, 2)}, %{n}")
        E.e(f"s_mov_b32 {sr(base + 3)}, 0x00020000")
    E.e(f"s_mov_b32 {sr(S['XD'] + 2)}, %4")
    E.e(f"s_mov_b32 {sr(S['CD'] + 2)}, %5")
    for n, name in enumerate(["WAVE", "NK", "LDA", "LDC", "KW", "FLAGS"]):
        E.e(f"s_mov_b32 {sr(S[name])}, %{6 + n}")
    E.e(f"s_lshl_b32 {sr(S['WD'] + 2)}, {sr(S['KW'])}, 8")              # 256 rows of W
    E.e(f"s_mov_b32 {sr(S['BD'] + 2)}, 512")
    E.e(f"s_lshl_b32 {sr(S['WDST'])}, {sr(S['WAVE'])}, 13")             # this wave's 8 pieces of a tile: 8 KiB
    E.e(f"s_add_u32 {sr(S['XDST'])}, {sr(S['WDST'])}, {XOFF}")
    E.e(f"s_mov_b32 {sr(S['KOFF'])}, 0")
    E.nops(4)


def emit_lane_setup(E):
    L, R, HH, SW, T0, T1, T2 = (V_T + i for i in range(7))
    E.e(f"v_mbcnt_lo_u32_b32 {vr(L)}, -1, 0")
    E.e(f"v_mbcnt_hi_u32_b32 {vr(L)}, -1, {vr(L)}")
    E.e(f"v_and_b32 {vr(R)}, 31, {vr(L)}")
    E.e(f"v_lshrrev_b32 {vr(HH)}, 5, {vr(L)}")
    # fragment read bases: (half*128 + r)*128 + (((2ks + hh) ^ ((r>>1)&7)) << 4), W half = wave>>1, X half = wave&1
    E.e(f"v_bfe_u32 {vr(SW)}, {vr(R)}, 1, 3")
    E.e(f"v_lshlrev_b32 {vr(T0)}, 7, {vr(R)}")
    E.e(f"s_lshr_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 1")
    E.e(f"s_lshl_b32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, 14")
    E.e(f"s_and_b32 {sr(S['TMP1'])}, {sr(S['WAVE'])}, 1")
    E.e(f"s_lshl_b32 {sr(S['TMP1'])}, {sr(S['TMP1'])}, 14")
    E.e(f"s_add_u32 {sr(S['TMP1'])}, {sr(S['TMP1'])}, {XOFF}")
    for ks in range(4):
        E.e(f"v_or_b32 {vr(T1)}, {2 * ks}, {vr(HH)}")
        E.e(f"v_xor_b32 {vr(T1)}, {vr(T1)}, {vr(SW)}")
        E.e(f"v_lshl_add_u32 {vr(T1)}, {vr(T1)}, 4, {vr(T0)}")
        E.e(f"v_add_u32 {vr(V_WRD + ks)}, {sr(S['TMP0'])}, {vr(T1)}")
        E.e(f"v_add_u32 {vr(V_XRD + ks)}, {sr(S['TMP1'])}, {vr(T1)}")
        E.e(f"v_add_u32 {vr(V_WRD + 4 + ks)}, {STAGE}, {vr(V_WRD + ks)}")
        E.e(f"v_add_u32 {vr(V_XRD + 4 + ks)}, {STAGE}, {vr(V_XRD + ks)}")
    # LDS-DMA source offsets: piece i of this wave covers tile rows 64w + 8i .. +7; lane: row = 64w + 8i + (l>>3),
    # source chunk = (l&7) ^ ((row>>1)&7) = (l&7) ^ ((4(i&1) + (l>>4)) & 7)
    E.e(f"v_lshrrev_b32 {vr(T0)}, 3, {vr(L)}")                          # l>>3
    E.e(f"s_lshl_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 6")
    E.e(f"v_add_u32 {vr(T0)}, {sr(S['TMP0'])}, {vr(T0)}")               # 64w + (l>>3)
    E.e(f"v_lshrrev_b32 {vr(T1)}, 4, {vr(L)}")                          # l>>4
    E.e(f"v_and_b32 {vr(T2)}, 7, {vr(L)}")
    for i in range(8):
        E.e(f"v_add_u32 {vr(SW)}, {4 * (i & 1)}, {vr(T1)}")
        E.e(f"v_and_b32 {vr(SW)}, 7, {vr(SW)}")
        E.e(f"v_xor_b32 {vr(SW)}, {vr(SW)}, {vr(T2)}")
        E.e(f"v_lshlrev_b32 {vr(SW)}, 4, {vr(SW)}")                     # source chunk * 16
        E.e(f"v_add_u32 {vr(R)}, {8 * i}, {vr(T0)}")                    # row
        E.e(f"v_mul_lo_u32 {vr(V_DW + i)}, {vr(R)}, {sr(S['KW'])}")
        E.e(f"v_add_u32 {vr(V_DW + i)}, {vr(V_DW + i)}, {vr(SW)}")
        E.e(f"v_mul_lo_u32 {vr(V_DX + i)}, {vr(R)}, {sr(S['LDA'])}")
        E.e(f"v_add_u32 {vr(V_DX + i)}, {vr(V_DX + i)}, {vr(SW)}")
    E.nops(2)


def dma_piece(op, i, stage, koff=None):
    base, dst, rs = (V_DW, S["WDST"], S["WD"]) if op == "W" else (V_DX, S["XDST"], S["XD"])
    koff = S["KOFF"] if koff is None else koff
    return [f"s_add_u32 m0, {sr(dst)}, {stage * STAGE + i * 1024}",
            "s_nop 0",
            f"buffer_load_dwordx4 {vr(base + i)}, {sr(rs, 4)}, {sr(koff)} offen lds"]


def frag_read(op, blk, fset, stage, ks):
    """ds_read_b128 of W fragment ni / X fragment mi of k-step ks into fragment set fset."""
    dst = fset + (0 if op == "W" else 16) + 4 * blk
    base = (V_WRD if op == "W" else V_XRD) + 4 * stage + ks
    return f"ds_read_b128 {vr(dst, 4)}, {vr(base)} offset:{blk * 4096}"


def mfma(ni, mi, fset):
    d = ar(acc(ni, mi), 16)
    return f"v_mfma_f32_32x32x16_bf16 {d}, {vr(fset + 4 * ni, 4)}, {vr(fset + 16 + 4 * mi, 4)}, {d}"


def build_iteration(E, stage, first, budget, koff_w=None, koff_x=None, advance=True):
    """One K-step of 64 (tile t in LDS stage `stage`): block 0 = last k-step of tile t-1 (fragment set F1, read before the
    barrier), blocks 1..3 = k-steps 0..2 of tile t; reads of k-step s+1 beside the MFMAs of k-step s; the LDS-DMA of tile t+1
    goes to the other stage.  first: the peeled first iteration (no block 0)."""
    items = []
    add = items.append
    sets = [F1, F0, F1, F0]                       # fragment set multiplied by block b
    # reads: R(t,0)->F0 (needed by block 1), R(t,1)->F1 (block 2; F1 busy in block 0), R(t,2)->F0 (block 3; busy in block 1),
    # R(t,3)->F1 (next iteration's block 0; busy in block 2)
    for ks, (fset, busy_blk, need_blk) in enumerate([(F0, None, 1), (F1, 0, 2), (F0, 1, 3), (F1, 2, 4)]):
        for op in ("W", "X"):
            for blk in range(4):
                if busy_blk is None or (first and busy_blk == 0):
                    earliest = 0
                else:      # W fragment ni is multiplied in gaps 16b + 4ni .. +3, X fragment mi last in gap 16b + 12 + mi
                    earliest = 16 * busy_blk + (4 * blk + 3 if op == "W" else 12 + blk) + 2
                need = 16 * need_blk + (4 * blk if op == "W" else blk)
                deadline = min(need - 4, 60)
                add(Item(f"rd{ks}{op}{blk}", [frag_read(op, blk, fset, stage, ks)], 2, earliest=earliest, deadline=max(deadline, earliest),
                         lds=1))
    for n, (op, i) in enumerate([("W", i) for i in range(8)] + [("X", i) for i in range(8)]):
        g0 = 1 + (n * DMA_EVERY_X2) // 2
        add(Item(f"dma{op}{i}", dma_piece(op, i, stage ^ 1, koff_w if op == "W" else koff_x), 12, earliest=g0, deadline=g0 + 8))
    gaps, load = schedule(items, 64, budget)
    lds_issued, lds_done, done_at = 0, 0, {}
    # reads of R(t-1,3) (set F1, block 0) completed before the barrier (lgkmcnt(0))
    for g in range(64):
        b, idx = g >> 4, g & 15
        ni, mi = idx >> 2, idx & 3
        fset = sets[b]
        if not (first and b == 0):
            ks = b - 1
            if b >= 1:
                need = max(done_at[f"rd{ks}W{ni}"], done_at[f"rd{ks}X{mi}"])
                if need > lds_done:
                    E.e(f"s_waitcnt lgkmcnt({min(lds_issued - need, 15)})")
                    lds_done = need if lds_issued - need <= 15 else lds_issued - 15
            E.e(mfma(ni, mi, fset))
        for it in gaps[g]:
            for ln in it.lines:
                E.e(ln)
            lds_issued += it.lds
            if it.lds:
                done_at[it.name] = lds_issued
    if advance:
        E.e(f"s_add_u32 {sr(S['KOFF'])}, {sr(S['KOFF'])}, 128")
    E.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    E.e("s_barrier")
    return load


def emit_epilogue(E):
    """acc -> + bias -> bf16 -> C.  Lane (r, hh) holds for (ni, mi, g): row m = 128*(wave&1) + 32mi + r, columns
    n = 128*(wave>>1) + 32ni + 8g + 4hh + 0..3."""
    L, R, HH, ROW, COL, T0 = (V_T + 8 + i for i in range(6))
    BIAS = 0                      # v0..63: bias of this lane's 16 (ni, g) column groups, unpacked to fp32
    E.nops(32)
    E.e(f"v_mbcnt_lo_u32_b32 {vr(L)}, -1, 0")
    E.e(f"v_mbcnt_hi_u32_b32 {vr(L)}, -1, {vr(L)}")
    E.e(f"v_and_b32 {vr(R)}, 31, {vr(L)}")
    E.e(f"v_lshrrev_b32 {vr(HH)}, 5, {vr(L)}")
    E.e(f"s_lshr_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 1")
    E.e(f"s_lshl_b32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, 8")                 # column byte offset of the wave: 128 cols * 2
    E.e(f"v_lshlrev_b32 {vr(COL)}, 3, {vr(HH)}")                           # 4hh cols * 2 bytes
    E.e(f"v_add_u32 {vr(COL)}, {sr(S['TMP0'])}, {vr(COL)}")
    for ni in range(4):
        for g in range(4):
            E.e(f"buffer_load_dwordx2 {vr(64 + 2 * (ni * 4 + g), 2)}, {vr(COL)}, {sr(S['BD'], 4)}, 0 offen offset:{64 * ni + 16 * g}")
    E.e(f"s_and_b32 {sr(S['TMP1'])}, {sr(S['WAVE'])}, 1")
    E.e(f"s_lshl_b32 {sr(S['TMP1'])}, {sr(S['TMP1'])}, 7")
    E.e(f"v_add_u32 {vr(ROW)}, {sr(S['TMP1'])}, {vr(R)}")
    E.e(f"v_mul_lo_u32 {vr(ROW)}, {vr(ROW)}, {sr(S['LDC'])}")
    E.e(f"v_add_u32 {vr(ROW)}, {vr(ROW)}, {vr(COL)}")                      # byte offset of (row, first column group) in the tile
    E.e(f"s_lshl_b32 {sr(S['TMP2'])}, {sr(S['LDC'])}, 5")                  # 32 rows
    E.e("s_waitcnt vmcnt(0)")
    for q in range(16):
        lo, hi = 64 + 2 * q, 65 + 2 * q
        E.e(f"v_and_b32 {vr(BIAS + 4 * q + 1)}, 0xffff0000, {vr(lo)}")
        E.e(f"v_lshlrev_b32 {vr(BIAS + 4 * q)}, 16, {vr(lo)}")
        E.e(f"v_and_b32 {vr(BIAS + 4 * q + 3)}, 0xffff0000, {vr(hi)}")
        E.e(f"v_lshlrev_b32 {vr(BIAS + 4 * q + 2)}, 16, {vr(hi)}")
    n = 0
    for mi in range(4):
        for ni in range(4):
            for g in range(4):
                tb = 64 + (n % 6) * 6                 # rotating temporaries v64..v99 (6 sets of 6)
                for j in range(4):
                    E.e(f"v_accvgpr_read_b32 {vr(tb + j)}, {ar(acc(ni, mi, 4 * g + j))}")
                E.e("s_nop 0")
                for j in range(4):
                    E.e(f"v_add_f32 {vr(tb + j)}, {vr(tb + j)}, {vr(BIAS + 4 * (ni * 4 + g) + j)}")
                E.e(f"v_cvt_pk_bf16_f32 {vr(tb + 4)}, {vr(tb)}, {vr(tb + 1)}")
                E.e(f"v_cvt_pk_bf16_f32 {vr(tb + 5)}, {vr(tb + 2)}, {vr(tb + 3)}")
                E.e(f"buffer_store_dwordx2 {vr(tb + 4, 2)}, {vr(ROW)}, {sr(S['CD'], 4)}, 0 offen offset:{64 * ni + 16 * g}")
                n += 1
                if n % 6 == 0:
                    E.e("s_waitcnt vmcnt(2)")
        if mi < 3:
            E.e(f"v_add_u32 {vr(ROW)}, {sr(S['TMP2'])}, {vr(ROW)}")
    E.e("s_waitcnt vmcnt(0)")


def generate(stamp, budget):
    E = Emitter()
    if stamp:
        E.e(f"s_memtime {sr(S['ST2'], 2)}")
    emit_inputs(E)
    emit_lane_setup(E)
    for a in range(256):
        E.e(f"v_accvgpr_write_b32 {ar(a)}, 0")
    for op in ("W", "X"):
        for i in range(8):
            for ln in dma_piece(op, i, 0):
                E.e(ln)
    E.e(f"s_add_u32 {sr(S['KOFF'])}, {sr(S['KOFF'])}, 128")
    E.e("s_waitcnt vmcnt(0)")
    E.e("s_barrier")
    if stamp:
        E.e(f"s_memtime {sr(S['ST0'], 2)}")
        E.e(f"s_memrealtime {sr(S['ST1'], 2)}")
        E.e("s_waitcnt lgkmcnt(0)")
    loop, done = E.label("gloop"), E.label("gdone")
    E.e(f"s_mov_b32 {sr(S['T'])}, 1")
    load = build_iteration(E, 0, True, budget)                    # tile 0 (peeled: no block 0)
    E.e(f"s_cmp_ge_u32 {sr(S['T'])}, {sr(S['NK'])}")
    E.e(f"s_cbranch_scc1 {done}")
    E.e(f"{loop}:")
    for stage in (1, 0):
        load = build_iteration(E, stage, False, budget)
        E.e(f"s_add_u32 {sr(S['T'])}, {sr(S['T'])}, 1")
        E.e(f"s_cmp_ge_u32 {sr(S['T'])}, {sr(S['NK'])}")
        if stage == 1:
            E.e(f"s_cbranch_scc1 {done}")
        else:
            E.e(f"s_cbranch_scc0 {loop}")
    E.e(f"{done}:")
    for idx in range(16):                                         # last k-step of the last tile
        E.e(mfma(idx >> 2, idx & 3, F1))
    if stamp:
        E.e(f"s_memtime {sr(S['TMP64'], 2)}")
        E.e(f"s_memrealtime {sr(S['BD'], 2)}")                   # (the bias descriptor is rebuilt below)
        E.e("s_waitcnt lgkmcnt(0)")
        E.e(f"s_sub_u32 {sr(S['ST0'])}, {sr(S['TMP64'])}, {sr(S['ST0'])}")
        E.e(f"s_sub_u32 {sr(S['ST1'])}, {sr(S['BD'])}, {sr(S['ST1'])}")
        E.e(f"s_mov_b64 {sr(S['BD'], 2)}, %3")
    emit_epilogue(E)
    if stamp:
        skip = E.label("nostamp")
        E.e(f"s_cmp_lg_u32 {sr(S['WAVE'])}, 0")
        E.e(f"s_cbranch_scc1 {skip}")
        E.e(f"s_memtime {sr(S['TMP64'], 2)}")
        E.e("s_waitcnt lgkmcnt(0)")
        E.e(f"s_sub_u32 {sr(S['TMP0'])}, {sr(S['TMP64'])}, {sr(S['ST2'])}")
        E.e(f"v_mov_b32 {vr(8)}, {sr(S['ST0'])}")
        E.e(f"v_mov_b32 {vr(9)}, {sr(S['ST1'])}")
        E.e(f"v_mov_b32 {vr(10)}, {sr(S['NK'])}")
        E.e(f"v_mov_b32 {vr(11)}, {sr(S['TMP0'])}")
        E.e(f"v_mov_b32 {vr(12)}, 0")
        E.e("s_mov_b64 exec, 1")
        E.e(f"buffer_store_dwordx4 {vr(8, 4)}, {vr(12)}, {sr(S['CD'], 4)}, 0 offen")
        E.e("s_waitcnt vmcnt(0)")
        E.e("s_mov_b64 exec, -1")
        E.e(f"{skip}:")
    return E, load


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stamp", action="store_true")
    ap.add_argument("--budget", type=int, default=18)
    ap.add_argument("--report", action="store_true")
    a = ap.parse_args()
    E, load = generate(a.stamp, a.budget)
    if a.report:
        print(f"gap loads {load} (max {max(load)}, sum {sum(load)})", file=sys.stderr)
    out = ["// GENERATED by gen_gemm_w4.py : do not edit", "#define FG_GEMM_W4_ASM \\"]
    for ln in E.lines:
        out.append('    "%s\\n\\t" \\' % ln)
    out.append('    ""')
    regs = [f'"v{i}"' for i in range(128)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(SB, SB + NSREG)]
    out.append("#define FG_GEMM_W4_CLOBBERS " + ", ".join(regs) + ', "vcc", "scc", "memory"')
    print("\n".join(out))


if __name__ == "__main__":
    main()
