#!/usr/bin/env python3
"""Generator of the hand-scheduled body of conv3d_cl_w4_kernel (csrc/vae_conv.hip): the implicit-GEMM causal convolution of the
VAE38 decoder on the 4-wave / 512-register GEMM core of gen_gemm_w4.py (256 couts x 256 pixels x 64 cin per step, LDS-DMA,
fragments read one k-step ahead across the barrier).

    python3 gen_conv_w4.py [--stamp] > conv_w4_asm.inc

What the convolution adds to the GEMM: the reduction runs over taps x cin; the pixel operand's rows are the output pixels
SHIFTED by the tap (zero padding = out-of-range buffer offsets), so the per-lane LDS-DMA source offsets change at every tap.
The wrapper (C++) tabulates, once per workgroup, the byte offset of every (tap, tile row) in LDS (table at 128 KiB:
off[tap][256 rows], then out[256 rows] for the epilogue); the body reloads its 8 piece offsets from that table when the tap
changes (8 ds_read_b32 every cin_pad/64 steps) and advances the weight offset by one tap.  Epilogue: + bias -> bf16 rounding
-> (+ residual) -> bf16, 8-byte channels-last stores, exactly the arithmetic of conv3d_cl_256_kernel.
"""
import argparse
import sys

from gen_attn_w4 import Emitter, vr, ar, sr
import gen_gemm_w4 as G

S = G.S
TAB = 131072                      # LDS byte address of the offset table (after the two 64 KiB stages)
OUT_TAB = TAB + 27 * 1024
# extra SGPRs (beyond gen_gemm_w4.S, which ends at SB+39)
X = {k: v + G.SB + 40 for k, v in dict(RD=0, KPT=4, NTAPS=5, WTAP=6, CSN=7, TAPN=8, WTB=9, KOFFW=10, KOFFX=11, NKTOT=12, HASRES=13).items()}
NSREG = 40 + 14
V_TAB = 100                       # 8 table addresses (this lane's row of each piece), advanced by 1 KiB per tap
V_CH = 108                        # 2: source chunk * 16 of even / odd pieces
V_TT = 110                        # 8 temporaries for the table reads


def emit_inputs(E):
    """%0 x base, %1 w tile base, %2 out base, %3 bias base, %4 residual base (64-bit each); %5 x bytes, %6 out bytes, %7 wave,
    %8 ksteps per tap, %9 ntaps, %10 weight tap stride (bytes), %11 packed-weight row bytes (cin_pad * 2), %12 has_residual, %13 bytes of packed weights from w tile base."""
    for n, base in enumerate((S["XD"], S["WD"], S["CD"], S["BD"], X["RD"])):
        E.e(f"s_mov_b64 {sr(base, 2)}, %{n}")
        E.e(f"s_mov_b32 {sr(base + 3)}, 0x00020000")
    E.e(f"s_mov_b32 {sr(S['XD'] + 2)}, %5")
    E.e(f"s_mov_b32 {sr(S['CD'] + 2)}, %6")
    E.e(f"s_mov_b32 {sr(X['RD'] + 2)}, %6")
    E.e(f"s_mov_b32 {sr(S['WAVE'])}, %7")
    E.e(f"s_mov_b32 {sr(X['KPT'])}, %8")
    E.e(f"s_mov_b32 {sr(X['NTAPS'])}, %9")
    E.e(f"s_mov_b32 {sr(X['WTAP'])}, %10")
    E.e(f"s_mov_b32 {sr(S['KW'])}, %11")
    E.e(f"s_mov_b32 {sr(X['HASRES'])}, %12")
    E.e(f"s_mul_i32 {sr(X['NKTOT'])}, {sr(X['KPT'])}, {sr(X['NTAPS'])}")
    # the tap / cin-step offset travels in the scalar offset, which the hardware range check ADDS to the vector offset
    # (measured: with num_records = one tap, taps >= 1 read back as zeros): num_records = the packed weights from this
    # cout tile's first row to their end, so the prefetch past the last tap reads zeros instead of faulting
    E.e(f"s_mov_b32 {sr(S['WD'] + 2)}, %13")
    E.e(f"s_mov_b32 {sr(S['BD'] + 2)}, 512")
    E.e(f"s_lshl_b32 {sr(S['WDST'])}, {sr(S['WAVE'])}, 13")
    E.e(f"s_add_u32 {sr(S['XDST'])}, {sr(S['WDST'])}, {G.XOFF}")
    E.e(f"s_mov_b32 {sr(S['LDA'])}, 0")                                  # (the GEMM lane setup multiplies rows by LDA: unused here)
    for name in ("CSN", "TAPN", "WTB", "KOFFW", "KOFFX"):
        E.e(f"s_mov_b32 {sr(X[name])}, 0")
    E.nops(4)


def emit_conv_lane_setup(E):
    """After G.emit_lane_setup: the pixel-side piece offsets come from the LDS table (row = 64w + 8i + (l>>3)); chunk*16 per parity."""
    L, T0, T1, T2 = (V_TT + i for i in range(4))
    E.e(f"v_mbcnt_lo_u32_b32 {vr(L)}, -1, 0")
    E.e(f"v_mbcnt_hi_u32_b32 {vr(L)}, -1, {vr(L)}")
    E.e(f"v_lshrrev_b32 {vr(T0)}, 3, {vr(L)}")
    E.e(f"s_lshl_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 6")
    E.e(f"v_add_u32 {vr(T0)}, {sr(S['TMP0'])}, {vr(T0)}")               # 64w + (l>>3)
    E.e(f"v_lshlrev_b32 {vr(T0)}, 2, {vr(T0)}")
    for i in range(8):
        E.e(f"v_add_u32 {vr(V_TAB + i)}, {TAB + 32 * i}, {vr(T0)}")     # row (64w + 8i + (l>>3)) * 4 bytes
    E.e(f"v_lshrrev_b32 {vr(T1)}, 4, {vr(L)}")
    E.e(f"v_and_b32 {vr(T2)}, 7, {vr(L)}")
    for par in range(2):
        E.e(f"v_add_u32 {vr(V_CH + par)}, {4 * par}, {vr(T1)}")
        E.e(f"v_and_b32 {vr(V_CH + par)}, 7, {vr(V_CH + par)}")
        E.e(f"v_xor_b32 {vr(V_CH + par)}, {vr(V_CH + par)}, {vr(T2)}")
        E.e(f"v_lshlrev_b32 {vr(V_CH + par)}, 4, {vr(V_CH + par)}")


def emit_load_tap_offsets(E):
    """V_DX[i] <- table[tap of the next tile][row_i] + chunk*16; the table addresses then move on to the following tap."""
    for i in range(8):
        E.e(f"ds_read_b32 {vr(V_TT + i)}, {vr(V_TAB + i)}")
    for i in range(8):
        E.e(f"v_add_u32 {vr(V_TAB + i)}, 1024, {vr(V_TAB + i)}")
    E.e("s_waitcnt lgkmcnt(0)")
    for i in range(8):
        E.e(f"v_add_u32 {vr(G.V_DX + i)}, {vr(V_TT + i)}, {vr(V_CH + (i & 1))}")


def emit_advance(E):
    """Counters of the tile whose LDS-DMA is issued in this iteration (the NEXT tile): cin step, tap, source offsets."""
    same = E.label("sametap")
    E.e(f"s_add_u32 {sr(X['CSN'])}, {sr(X['CSN'])}, 1")
    E.e(f"s_cmp_lg_u32 {sr(X['CSN'])}, {sr(X['KPT'])}")
    E.e(f"s_cbranch_scc1 {same}")
    E.e(f"s_mov_b32 {sr(X['CSN'])}, 0")
    E.e(f"s_add_u32 {sr(X['WTB'])}, {sr(X['WTB'])}, {sr(X['WTAP'])}")
    emit_load_tap_offsets(E)
    E.e(f"{same}:")
    E.e(f"s_lshl_b32 {sr(X['KOFFX'])}, {sr(X['CSN'])}, 7")
    E.e(f"s_add_u32 {sr(X['KOFFW'])}, {sr(X['WTB'])}, {sr(X['KOFFX'])}")


def emit_epilogue(E):
    """acc + bias -> bf16 -> (+ residual) -> bf16 -> out.  Lane (r, hh) holds for (ni, mi, g): pixel row 128*(wave&1) + 32mi + r of
    the tile, couts 128*(wave>>1) + 32ni + 8g + 4hh + 0..3."""
    L, R, HH, COL, T0 = (G.V_T + 24 + i for i in range(5))            # v120..124
    OUTOFF = G.V_T + 20                                               # v116..119: byte offset of this lane's 4 pixels
    BIAS = 0
    E.nops(32)
    E.e(f"v_mbcnt_lo_u32_b32 {vr(L)}, -1, 0")
    E.e(f"v_mbcnt_hi_u32_b32 {vr(L)}, -1, {vr(L)}")
    E.e(f"v_and_b32 {vr(R)}, 31, {vr(L)}")
    E.e(f"v_lshrrev_b32 {vr(HH)}, 5, {vr(L)}")
    E.e(f"s_lshr_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 1")
    E.e(f"s_lshl_b32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, 8")
    E.e(f"v_lshlrev_b32 {vr(COL)}, 3, {vr(HH)}")
    E.e(f"v_add_u32 {vr(COL)}, {sr(S['TMP0'])}, {vr(COL)}")             # cout byte offset inside the 256-cout tile
    for ni in range(4):
        for g in range(4):
            E.e(f"buffer_load_dwordx2 {vr(64 + 2 * (ni * 4 + g), 2)}, {vr(COL)}, {sr(S['BD'], 4)}, 0 offen offset:{64 * ni + 16 * g}")
    E.e(f"s_and_b32 {sr(S['TMP1'])}, {sr(S['WAVE'])}, 1")
    E.e(f"s_lshl_b32 {sr(S['TMP1'])}, {sr(S['TMP1'])}, 9")              # 128 rows * 4 bytes
    E.e(f"v_lshlrev_b32 {vr(T0)}, 2, {vr(R)}")
    E.e(f"v_add_u32 {vr(T0)}, {sr(S['TMP1'])}, {vr(T0)}")
    E.e(f"v_add_u32 {vr(T0)}, {OUT_TAB}, {vr(T0)}")
    for mi in range(4):
        E.e(f"ds_read_b32 {vr(OUTOFF + mi)}, {vr(T0)} offset:{128 * mi}")
    E.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    for mi in range(4):
        E.e(f"v_add_u32 {vr(OUTOFF + mi)}, {vr(OUTOFF + mi)}, {vr(COL)}")
    for q in range(16):
        lo, hi = 64 + 2 * q, 65 + 2 * q
        E.e(f"v_and_b32 {vr(BIAS + 4 * q + 1)}, 0xffff0000, {vr(lo)}")
        E.e(f"v_lshlrev_b32 {vr(BIAS + 4 * q)}, 16, {vr(lo)}")
        E.e(f"v_and_b32 {vr(BIAS + 4 * q + 3)}, 0xffff0000, {vr(hi)}")
        E.e(f"v_lshlrev_b32 {vr(BIAS + 4 * q + 2)}, 16, {vr(hi)}")
    nores, done = E.label("nores"), E.label("epidone")
    E.e(f"s_cmp_eq_u32 {sr(X['HASRES'])}, 0")
    E.e(f"s_cbranch_scc1 {nores}")
    # ---- with residual: per pixel row, the 16 residual words of this lane first (32 VGPRs v64..95), then the groups
    for mi in range(4):
        for ni in range(4):
            for g in range(4):
                E.e(f"buffer_load_dwordx2 {vr(64 + 2 * (ni * 4 + g), 2)}, {vr(OUTOFF + mi)}, {sr(X['RD'], 4)}, 0 offen offset:{64 * ni + 16 * g}")
        E.e("s_waitcnt vmcnt(0)")
        for ni in range(4):
            for g in range(4):
                q = ni * 4 + g
                tb = 96 + (q % 2) * 10                # two temporary sets v96..115
                for j in range(4):
                    E.e(f"v_accvgpr_read_b32 {vr(tb + j)}, {ar(G.acc(ni, mi, 4 * g + j))}")
                E.e("s_nop 0")
                for j in range(4):
                    E.e(f"v_add_f32 {vr(tb + j)}, {vr(tb + j)}, {vr(BIAS + 4 * q + j)}")
                E.e(f"v_cvt_pk_bf16_f32 {vr(tb + 4)}, {vr(tb)}, {vr(tb + 1)}")           # rbf(acc + bias)
                E.e(f"v_cvt_pk_bf16_f32 {vr(tb + 5)}, {vr(tb + 2)}, {vr(tb + 3)}")
                for w, src in ((0, tb + 4), (1, tb + 5)):
                    rw = 64 + 2 * q + w
                    E.e(f"v_lshlrev_b32 {vr(tb)}, 16, {vr(src)}")
                    E.e(f"v_and_b32 {vr(tb + 1)}, 0xffff0000, {vr(src)}")
                    E.e(f"v_lshlrev_b32 {vr(tb + 2)}, 16, {vr(rw)}")
                    E.e(f"v_and_b32 {vr(tb + 3)}, 0xffff0000, {vr(rw)}")
                    E.e(f"v_add_f32 {vr(tb)}, {vr(tb)}, {vr(tb + 2)}")
                    E.e(f"v_add_f32 {vr(tb + 1)}, {vr(tb + 1)}, {vr(tb + 3)}")
                    E.e(f"v_cvt_pk_bf16_f32 {vr(tb + 6 + w)}, {vr(tb)}, {vr(tb + 1)}")
                E.e(f"buffer_store_dwordx2 {vr(tb + 6, 2)}, {vr(OUTOFF + mi)}, {sr(S['CD'], 4)}, 0 offen offset:{64 * ni + 16 * g}")
                if q % 2 == 1:
                    E.e("s_waitcnt vmcnt(1)")
        E.e("s_waitcnt vmcnt(0)")
    E.e(f"s_branch {done}")
    # ---- without residual
    E.e(f"{nores}:")
    n = 0
    for mi in range(4):
        for ni in range(4):
            for g in range(4):
                tb = 64 + (n % 6) * 6
                for j in range(4):
                    E.e(f"v_accvgpr_read_b32 {vr(tb + j)}, {ar(G.acc(ni, mi, 4 * g + j))}")
                E.e("s_nop 0")
                for j in range(4):
                    E.e(f"v_add_f32 {vr(tb + j)}, {vr(tb + j)}, {vr(BIAS + 4 * (ni * 4 + g) + j)}")
                E.e(f"v_cvt_pk_bf16_f32 {vr(tb + 4)}, {vr(tb)}, {vr(tb + 1)}")
                E.e(f"v_cvt_pk_bf16_f32 {vr(tb + 5)}, {vr(tb + 2)}, {vr(tb + 3)}")
                E.e(f"buffer_store_dwordx2 {vr(tb + 4, 2)}, {vr(OUTOFF + mi)}, {sr(S['CD'], 4)}, 0 offen offset:{64 * ni + 16 * g}")
                n += 1
                if n % 6 == 0:
                    E.e("s_waitcnt vmcnt(2)")
    E.e(f"{done}:")
    E.e("s_waitcnt vmcnt(0)")


def generate(stamp, budget):
    E = Emitter()
    if stamp:
        E.e(f"s_memtime {sr(S['ST2'], 2)}")
    emit_inputs(E)
    G.emit_lane_setup(E)
    emit_conv_lane_setup(E)
    for a in range(256):
        E.e(f"v_accvgpr_write_b32 {ar(a)}, 0")
    # tile 0 = (tap 0, cin step 0)
    emit_load_tap_offsets(E)
    for op in ("W", "X"):
        for i in range(8):
            for ln in G.dma_piece(op, i, 0, X["KOFFW"] if op == "W" else X["KOFFX"]):
                E.e(ln)
    E.e("s_waitcnt vmcnt(0)")
    E.e("s_barrier")
    if stamp:
        E.e(f"s_memtime {sr(S['ST0'], 2)}")
        E.e(f"s_memrealtime {sr(S['ST1'], 2)}")
        E.e("s_waitcnt lgkmcnt(0)")
    loop, done = E.label("cloop"), E.label("cdone")
    E.e(f"s_mov_b32 {sr(S['T'])}, 1")
    emit_advance(E)
    load = G.build_iteration(E, 0, True, budget, X["KOFFW"], X["KOFFX"], advance=False)
    E.e(f"s_cmp_ge_u32 {sr(S['T'])}, {sr(X['NKTOT'])}")
    E.e(f"s_cbranch_scc1 {done}")
    E.e(f"{loop}:")
    for stage in (1, 0):
        emit_advance(E)
        load = G.build_iteration(E, stage, False, budget, X["KOFFW"], X["KOFFX"], advance=False)
        E.e(f"s_add_u32 {sr(S['T'])}, {sr(S['T'])}, 1")
        E.e(f"s_cmp_ge_u32 {sr(S['T'])}, {sr(X['NKTOT'])}")
        if stage == 1:
            E.e(f"s_cbranch_scc1 {done}")
        else:
            E.e(f"s_cbranch_scc0 {loop}")
    E.e(f"{done}:")
    for idx in range(16):
        E.e(G.mfma(idx >> 2, idx & 3, G.F1))
    if stamp:
        E.e(f"s_memtime {sr(S['TMP64'], 2)}")
        E.e(f"s_memrealtime {sr(S['ST2'], 2)}")
        E.e("s_waitcnt lgkmcnt(0)")
        E.e(f"s_sub_u32 {sr(S['ST0'])}, {sr(S['TMP64'])}, {sr(S['ST0'])}")
        E.e(f"s_sub_u32 {sr(S['ST1'])}, {sr(S['ST2'])}, {sr(S['ST1'])}")
    emit_epilogue(E)
    if stamp:       # every workgroup's wave 0 overwrites the first 16 bytes of its first pixel row (cout tile 0 only is decoded)
        skip = E.label("nostamp")
        E.e(f"s_cmp_lg_u32 {sr(S['WAVE'])}, 0")
        E.e(f"s_cbranch_scc1 {skip}")
        E.e(f"v_mov_b32 {vr(8)}, {sr(S['ST0'])}")
        E.e(f"v_mov_b32 {vr(9)}, {sr(S['ST1'])}")
        E.e(f"v_mov_b32 {vr(10)}, {sr(X['NKTOT'])}")
        E.e(f"v_mov_b32 {vr(11)}, 0")
        E.e(f"v_mov_b32 {vr(12)}, {OUT_TAB}")
        E.e(f"ds_read_b32 {vr(12)}, {vr(12)}")
        E.e("s_waitcnt lgkmcnt(0)")
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
    ap.add_argument("--dma-every-x2", type=int, default=G.DMA_EVERY_X2)
    a = ap.parse_args()
    G.DMA_EVERY_X2 = a.dma_every_x2
    E, load = generate(a.stamp, a.budget)
    out = ["// GENERATED by gen_conv_w4.py : do not edit", "#define FG_CONV_W4_ASM \\"]
    for ln in E.lines:
        out.append('    "%s\\n\\t" \\' % ln)
    out.append('    ""')
    regs = [f'"v{i}"' for i in range(128)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(G.SB, G.SB + NSREG)]
    out.append("#define FG_CONV_W4_CLOBBERS " + ", ".join(regs) + ', "vcc", "scc", "memory"')
    print("\n".join(out))


if __name__ == "__main__":
    main()
