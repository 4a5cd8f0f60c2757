#!/usr/bin/env python3
"""Generator of the hand-scheduled body of gemm_p_kernel (csrc/dit_gemm.hip), the persistent DiT GEMM behind fg_gemm_epilogue_bf16:
C[M,N] = A[M,K] W[N,K]^T + bias (+ residual epilogue), bf16 in / out, fp32 accumulate (nn.Linear / GateModule of the DiT block,
models/wan_video_dit.py:130-133,188-193,208-209,225-228).

    python3 gen_gemm_p.py --nb 4 --tail 1 | --nb 3  [--stamp] [--pf-dist N] [--ablate nodma|samek|l1]      (run by the Makefile)

Core: 4 waves, one per SIMD, all 512 registers owned by the asm; 256 activation rows x NB*64 weight rows x 64 k per step; operand
tiles by LDS-DMA into a ring of LDS stages; fragments read one k-step ahead, across the barrier; every fragment read feeds NB or
4 MFMAs.  Around it (DESIGN.md §5 has the measurements behind each point):
  * ONE workgroup per CU walks a list of 64-byte tile records the C++ wrapper wrote to LDS (operand / output / bias bases, valid
    bytes, k range, gate slice, prefetch lanes); the next tile's record and first k-slice are fetched BEFORE the current tile's
    epilogue, the bias words when a tile starts; the first k-step multiplies onto C = 0;
  * cooperative L2 prefetch: the CUs of an XCD that share an operand slice each pull a part of its rows pf_dist slices ahead with
    EXEC-masked dummy loads, and the step's wait is counted so the prefetch never blocks;
  * epilogue through a per-wave LDS transpose: every global access is a dwordx4 of 4 rows x 256 contiguous bytes; FLAGS 0 = bias,
    2 = x + gate * bf16(acc + bias) (x read in place, gate per token class), 3 = x + bf16(acc + bias), 4 = bf16(gelu_tanh(bf16(acc +
    bias))); KIND 2 records (k-range
    pieces of the last round's tiles) store their fp32 accumulators raw for gemm_reduce_kernel;
  * --tail 1: a second body for 256 x 64 column pieces of the last round's tiles, on a 3-stage ring (its shorter steps would be
    DMA-latency-bound on 2 stages); --nb 3: the 256 x 192 tile for N % 256 != 0.
--stamp: in-kernel cycle counters per phase (tools/gemm_ab.py --stamp-lib-p); --ablate: timing-only builds (wrong results) used to
locate the k-step's cost; neither is part of the product library.
"""
import argparse
import sys

from gen_attn_w4 import Emitter, Item, schedule, vr, ar, sr

STAGE, XOFF = 65536, 32768
TAB = 131072                           # tile records (48 B each) above the two stages
SB = 36                                # s32..s34 are reserved by the compiler (stack / frame / base pointer)
S = {k: v + SB for k, v in dict(XD=0, WD=4, CD=8, BD=12, GD=16, WAVE=20, NK=21, XHOME=22, LDC=23, KOFF=24, T=25, UNIT=26, TMP0=27,
                                TMP1=28, TMP2=29, WDST=30, XDST=31, NTILES=32, FLAGS=33, FIRST=34, GLD=35, TMP64=36, ST1=38,
                                KARG=40, M0ROW=42, KIND=43, REC=44).items()}      # REC: 16 SGPRs of the current tile record
# The scheduling / shape block at the start of the kernel arguments (struct GemmSched in dit_gemm.hip, same offsets): the body reads it
# with scalar loads through KARG
K_A, K_W, K_C, K_BIAS, K_WS, K_CURS, K_GATE, K_SCALE = 0, 8, 16, 24, 32, 40, 48, 56
K_DIMS0 = 64       # M, lda bytes, ldc bytes, k bytes
K_DIMS1 = 80       # nk, tiles_m, tiles_n, gm | gm * tiles_n, its division magic, nxcd, grid
K_QD = 112         # qd, rm, per, flags
K_EPI = 128        # first_rows, gate ld bytes, gate bytes, M - 1
K_PLAN = 144       # two plans of 8 dwords (XCDs with qd + 1 / qd tiles): n_whole, n_main, n_tail, split, split magic, base, extra, full * per
PIECE_BYTES = 256 * 256 * 4
NSREG = 62                             # ... and the L2-prefetch lane shifts (s96, s97)
PF_SX, PF_SW = SB + 60, SB + 61
F0, F1 = 0, 32
V_WRD, V_XRD = 64, 72
V_DW, V_DX = 80, 88
# lane constants of the epilogue (set up once per kernel)
V_COLB = 96        # fragment layout (lane = (r, hh): row r of a 32-row block, columns 8g + 4hh + 0..3): column byte offset in the tile
V_LW = 97          # LDS address this lane writes its fragment-layout words to (+ 64ni + 16g)
V_LR = 98          # LDS address this lane reads row-major words from: lane i = row i>>4 of a 4-row group, columns 8*(i&15) + 0..7
V_SOFF = 99        # row-major byte offset in the output tile (out of range for the idle lanes of the 192-column tile)
V_RROW = 100       # row-major row of the tile, without the 32mi + 4it part
V_GOFF = 101       # row-major column byte offset (gate table)
V_TB = 102         # v102..111 temporaries
V_BIASRAW = 112    # v112..143: the tile's bias words, loaded when the tile starts (they land during the k-loop)
V_YR = 144         # v144..175: one 32-row block of y read back row-major (8 x 4 words)
V_XA, V_XB = 176, 208      # residual words of the current / next 32-row block
V_G0, V_G1 = 240, 244      # the lane's 8 gate values for token class 0 / 1 (loaded when the tile starts)
V_PW, V_PX, V_PD = 248, 249, 250   # L2 prefetch: byte offset of this lane's row of the wave's W / X tile part (one 128-byte line per
                                   # lane and k-slice), dummy destination
V_T = 224          # setup temporaries v224..230
V_REC = 144        # v144, v146..161: tile record loads (outside the epilogue)
V_SC = 252         # fp8 form only: v252..255 = the activation row scales of the lane's four 32-row blocks (loaded when the tile starts)
FP8 = False        # --dtype fp8: e4m3 operands (one byte per element, 128 k per LDS row), v_mfma_f32_32x32x64_f8f6f4, row-scale epilogue


def configure_fp8():
    """Register map of the e4m3 form.  A fragment is 8 registers (32 bytes: k = 64 s + 32 hh + 0..31 of the 128-k LDS row, s the k-substep),
    so the two fragment sets take v0..127; everything the epilogue alone uses (unpacked bias, y, the first residual buffer, record loads,
    setup temporaries) lives in that area, dead outside the k-loop."""
    globals().update(F0=0, F1=64, V_WRD=128, V_XRD=136, V_DW=144, V_DX=152, V_COLB=160, V_LW=161, V_LR=162, V_SOFF=163, V_RROW=164,
                     V_GOFF=165, V_TB=166, V_BIASRAW=176, V_YR=64, V_XA=96, V_XB=208, V_G0=240, V_G1=244, V_PW=248, V_PX=249,
                     V_PD=250, V_T=208, V_REC=64, V_SC=252, FP8=True)
    globals().update(V_GC=96, V_GT=104)


OOB = 0x80000000


class Cfg:
    dma_step = 2.0                       # MFMA gaps between two LDS-DMA pieces (--dma-step)
    dma_first = 1                        # gap of the first piece
    serp = 1                             # MFMA order inside a k-substep: 1 = boustrophedon over the X fragments: consecutive MFMAs always share an
                                         # input operand (+0.5 % clock at the power cap, measured); 0 = row-major
    pf_coop = 1                          # 1: the CUs of an XCD that share an operand slice pull a part of its rows each (EXEC-masked): every
                                         #    8th activation row (8 tiles of a row block), every 4th weight row; 2: activation rows only
    pf_dist = 2                          # L2 prefetch distance in k-slices beyond the slice the step's LDS-DMA fetches (0 = off)
    ablate = ""                          # timing-only builds (wrong results): "nodma" = no LDS-DMA inside the k-loop, "samek", "l1"

    def __init__(self, nb):
        self.nb = nb                     # 32-row weight blocks per wave
        self.g16 = nb * 4                # MFMAs per k-substep
        self.nsub = 2 if FP8 else 4      # k-substeps (MFMA K) per LDS stage of 128 bytes per row: 4 x 16 bf16 / 2 x 64 e4m3
        self.fr = 8 if FP8 else 4        # registers per fragment
        self.gaps = self.nsub * self.g16 # MFMAs per stage
        self.npw = 2 * nb                # weight LDS-DMA pieces per wave per tile
        # LDS ring of operand stages.  256-/192-column tiles: 2 stages of 64 KiB.  The 64-column tail tile moves 40 KiB per k-step
        # for a quarter of the MFMAs, so with 2 stages its step would last one LDS-DMA round trip (measured: ~2 000 cycles, as long
        # as a full tile's); its 40 KiB stages fit a ring of 3 and the DMA of slice t+2 has two steps to land.
        self.ring = 3 if nb == 1 else 2
        self.stage_bytes = 40960 if nb == 1 else STAGE
        self.xoff = 8192 if nb == 1 else XOFF
        # the epilogue's transpose buffers: stage 1 (2 stages: the next tile's first slice goes to stage 0) / above the tile table
        self.lbase = STAGE if self.ring == 2 else TAB + 8192
        self.rs = nb * 64 + 16           # LDS row stride of the transpose buffer (bytes): a 32-row block of the wave's 128 x NB*32 part
        self.wsz = 32 * self.rs


def acc(ni, mi, j=0):
    return (ni * 4 + mi) * 16 + j


def emit_setup(E, C, first=True):
    """Asm operands: %0 wave, %1 kernel argument segment (64-bit; struct GemmSched at its start), %2 the workgroup's home XCD (blockIdx %
    nxcd).  Everything else comes from the argument segment.  LDA / KW: operand row strides in bytes (REC is scratch here)."""
    if first:
        E.e(f"s_mov_b32 {sr(S['WAVE'])}, %0")
        E.e(f"s_mov_b64 {sr(S['KARG'], 2)}, %1")
        E.e(f"s_mov_b32 {sr(S['XHOME'])}, %2")
        E.e(f"s_mov_b32 {sr(S['NTILES'])}, 0")
    E.e(f"s_load_dwordx4 {sr(S['REC'], 4)}, {sr(S['KARG'], 2)}, {K_DIMS0}")
    E.e(f"s_load_dwordx4 {sr(S['REC'] + 4, 4)}, {sr(S['KARG'], 2)}, {K_QD}")
    E.e(f"s_load_dwordx4 {sr(S['REC'] + 8, 4)}, {sr(S['KARG'], 2)}, {K_EPI}")
    E.e("s_waitcnt lgkmcnt(0)")
    LDA, KW = S["REC"] + 1, S["REC"] + 3
    E.e(f"s_mov_b32 {sr(S['LDC'])}, {sr(S['REC'] + 2)}")
    E.e(f"s_mov_b32 {sr(S['FLAGS'])}, {sr(S['REC'] + 7)}")
    E.e(f"s_mov_b32 {sr(S['FIRST'])}, {sr(S['REC'] + 8)}")
    E.e(f"s_mov_b32 {sr(S['GLD'])}, {sr(S['REC'] + 9)}")
    E.e(f"s_mov_b32 {sr(S['GD'] + 2)}, {sr(S['REC'] + 10)}")
    if FP8:
        E.e(f"s_mov_b32 {sr(S_MLAST)}, {sr(S['REC'] + 11)}")
        E.e(f"s_load_dwordx2 {sr(S_SCB, 2)}, {sr(S['KARG'], 2)}, {K_SCALE}")
    for base in (S["XD"], S["WD"], S["CD"], S["BD"], S["GD"]):
        E.e(f"s_mov_b32 {sr(base + 3)}, 0x00020000")
    E.e(f"s_mul_i32 {sr(S['WDST'])}, {sr(S['WAVE'])}, {C.npw * 1024}")
    E.e(f"s_lshl_b32 {sr(S['XDST'])}, {sr(S['WAVE'])}, 13")
    E.e(f"s_add_u32 {sr(S['XDST'])}, {sr(S['XDST'])}, {C.xoff}")
    E.e(f"v_mov_b32 {vr(255)}, 0")
    E.nops(4)
    L, R, HH, SW, T0, T1, T2 = (V_T + i for i in range(7))
    E.e(f"v_mbcnt_lo_u32_b32 {vr(L)}, -1, 0")
    E.e(f"v_mbcnt_hi_u32_b32 {vr(L)}, -1, {vr(L)}")
    E.e(f"v_and_b32 {vr(R)}, 31, {vr(L)}")
    E.e(f"v_lshrrev_b32 {vr(HH)}, 5, {vr(L)}")
    E.e(f"v_bfe_u32 {vr(SW)}, {vr(R)}, 1, 3")
    E.e(f"v_lshlrev_b32 {vr(T0)}, 7, {vr(R)}")
    E.e(f"s_lshr_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 1")
    E.e(f"s_mul_i32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, {C.nb * 4096}")          # weight half of this wave
    E.e(f"s_and_b32 {sr(S['TMP1'])}, {sr(S['WAVE'])}, 1")
    E.e(f"s_lshl_b32 {sr(S['TMP1'])}, {sr(S['TMP1'])}, 14")
    E.e(f"s_add_u32 {sr(S['TMP1'])}, {sr(S['TMP1'])}, {C.xoff}")
    for ks in range(4):
        if FP8:      # address register 2s + c: 16-byte chunk 4s + 2hh + c of the row (k-substep s, half c of the lane's 32 bytes)
            E.e(f"v_lshlrev_b32 {vr(T1)}, 1, {vr(HH)}")
            E.e(f"v_or_b32 {vr(T1)}, {4 * (ks >> 1) + (ks & 1)}, {vr(T1)}")
        else:
            E.e(f"v_or_b32 {vr(T1)}, {2 * ks}, {vr(HH)}")
        E.e(f"v_xor_b32 {vr(T1)}, {vr(T1)}, {vr(SW)}")
        E.e(f"v_lshl_add_u32 {vr(T1)}, {vr(T1)}, 4, {vr(T0)}")
        E.e(f"v_add_u32 {vr(V_WRD + ks)}, {sr(S['TMP0'])}, {vr(T1)}")
        E.e(f"v_add_u32 {vr(V_XRD + ks)}, {sr(S['TMP1'])}, {vr(T1)}")
        second = C.stage_bytes * (2 if C.ring == 3 else 1)      # ring of 3: set 0 serves stages 0 and 1 (immediate offset), set 1 stage 2
        E.e(f"v_add_u32 {vr(V_WRD + 4 + ks)}, {second}, {vr(V_WRD + ks)}")
        E.e(f"v_add_u32 {vr(V_XRD + 4 + ks)}, {second}, {vr(V_XRD + ks)}")
    # LDS-DMA source offsets (piece i of this wave: tile rows 8*(w*P + i) .. +7)
    E.e(f"v_lshrrev_b32 {vr(T0)}, 3, {vr(L)}")
    E.e(f"v_lshrrev_b32 {vr(T1)}, 4, {vr(L)}")
    E.e(f"v_and_b32 {vr(T2)}, 7, {vr(L)}")
    for op, npw, base, stride in (("W", C.npw, V_DW, KW), ("X", 8, V_DX, LDA)):
        E.e(f"s_mul_i32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, {8 * npw}")
        for i in range(npw):
            E.e(f"v_add_u32 {vr(SW)}, {4 * (i & 1)}, {vr(T1)}")
            E.e(f"v_and_b32 {vr(SW)}, 7, {vr(SW)}")
            E.e(f"v_xor_b32 {vr(SW)}, {vr(SW)}, {vr(T2)}")
            E.e(f"v_lshlrev_b32 {vr(SW)}, 4, {vr(SW)}")
            E.e(f"v_add_u32 {vr(R)}, {8 * i}, {vr(T0)}")
            E.e(f"v_add_u32 {vr(R)}, {sr(S['TMP0'])}, {vr(R)}")
            E.e(f"v_mul_lo_u32 {vr(base + i)}, {vr(R)}, {sr(stride)}")
            E.e(f"v_add_u32 {vr(base + i)}, {vr(base + i)}, {vr(SW)}")
    # L2 prefetch offsets: lane l <-> row l of this wave's 8*npw weight rows / 64 activation rows
    E.e(f"s_mul_i32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, {8 * C.npw}")
    E.e(f"v_add_u32 {vr(V_PW)}, {sr(S['TMP0'])}, {vr(L)}")
    E.e(f"v_mul_lo_u32 {vr(V_PW)}, {vr(V_PW)}, {sr(KW)}")
    if 8 * C.npw < 64:
        E.e(f"v_cmp_le_u32 vcc, {8 * C.npw}, {vr(L)}")
        E.e(f"v_mov_b32 {vr(T2)}, {OOB}")
        E.e("s_nop 1")
        E.e(f"v_cndmask_b32 {vr(V_PW)}, {vr(V_PW)}, {vr(T2)}, vcc")
    E.e(f"s_lshl_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 6")
    E.e(f"v_add_u32 {vr(V_PX)}, {sr(S['TMP0'])}, {vr(L)}")
    E.e(f"v_mul_lo_u32 {vr(V_PX)}, {vr(V_PX)}, {sr(LDA)}")
    # epilogue lane constants (see the register map)
    I4, I15 = T0, T1
    E.e(f"v_and_b32 {vr(R)}, 31, {vr(L)}")
    E.e(f"v_lshrrev_b32 {vr(HH)}, 5, {vr(L)}")
    E.e(f"v_lshrrev_b32 {vr(I4)}, 4, {vr(L)}")
    E.e(f"v_and_b32 {vr(I15)}, 15, {vr(L)}")
    E.e(f"s_lshr_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 1")
    E.e(f"s_mul_i32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, {C.nb * 64}")            # column byte offset of the wave's part
    E.e(f"v_lshlrev_b32 {vr(V_COLB)}, 3, {vr(HH)}")
    E.e(f"v_add_u32 {vr(V_COLB)}, {sr(S['TMP0'])}, {vr(V_COLB)}")
    E.e(f"s_mul_i32 {sr(S['TMP1'])}, {sr(S['WAVE'])}, {C.wsz}")
    E.e(f"s_add_u32 {sr(S['TMP1'])}, {sr(S['TMP1'])}, {C.lbase}")             # this wave's transpose buffer
    E.e(f"v_mul_u32_u24 {vr(V_LW)}, {C.rs}, {vr(R)}")
    E.e(f"v_lshl_add_u32 {vr(V_LW)}, {vr(HH)}, 3, {vr(V_LW)}")
    E.e(f"v_add_u32 {vr(V_LW)}, {sr(S['TMP1'])}, {vr(V_LW)}")
    E.e(f"v_mul_u32_u24 {vr(V_LR)}, {C.rs}, {vr(I4)}")
    E.e(f"v_lshl_add_u32 {vr(V_LR)}, {vr(I15)}, 4, {vr(V_LR)}")
    E.e(f"v_add_u32 {vr(V_LR)}, {sr(S['TMP1'])}, {vr(V_LR)}")
    E.e(f"s_and_b32 {sr(S['TMP1'])}, {sr(S['WAVE'])}, 1")
    E.e(f"s_lshl_b32 {sr(S['TMP1'])}, {sr(S['TMP1'])}, 7")
    E.e(f"v_add_u32 {vr(V_RROW)}, {sr(S['TMP1'])}, {vr(I4)}")
    E.e(f"v_lshlrev_b32 {vr(V_GOFF)}, 4, {vr(I15)}")
    E.e(f"v_add_u32 {vr(V_GOFF)}, {sr(S['TMP0'])}, {vr(V_GOFF)}")
    if C.nb < 4:       # lanes 4*NB..15 of every 16 have no columns: their global accesses go out of range (loads 0, stores dropped)
        E.e(f"v_cmp_le_u32 vcc, {4 * C.nb}, {vr(I15)}")
        E.e(f"v_mov_b32 {vr(T2)}, {OOB}")
        E.e("s_nop 1")
        E.e(f"v_cndmask_b32 {vr(V_GOFF)}, {vr(V_GOFF)}, {vr(T2)}, vcc")
    E.e(f"v_mul_lo_u32 {vr(V_SOFF)}, {vr(V_RROW)}, {sr(S['LDC'])}")
    E.e(f"v_add_u32 {vr(V_SOFF)}, {vr(V_SOFF)}, {vr(V_GOFF)}")
    if C.nb < 4:
        E.e(f"v_cndmask_b32 {vr(V_SOFF)}, {vr(V_SOFF)}, {vr(T2)}, vcc")
    E.e("s_waitcnt lgkmcnt(0)")
    E.nops(2)


ACC_K, ACC_E, ACC_W, T_START, R_START = S["ST1"] + 1, SB + 62, SB + 63, SB + 64, SB + 65     # --stamp only (s98..s101)


def emit_stamp(E, acc):
    """acc += cycles since the previous stamp (s_memtime, low 32 bits)."""
    E.e(f"s_memtime {sr(S['TMP64'], 2)}")
    E.e("s_waitcnt lgkmcnt(0)")
    E.e(f"s_sub_u32 {sr(S['TMP2'])}, {sr(S['TMP64'])}, {sr(S['ST1'])}")
    E.e(f"s_add_u32 {sr(acc)}, {sr(acc)}, {sr(S['TMP2'])}")
    E.e(f"s_mov_b32 {sr(S['ST1'])}, {sr(S['TMP64'])}")


V_GRAB = lambda: V_TB + 9       # noqa: E731  wave 0, lane 0: the unit index the cursor returned (in flight during the k-loop)
V_SLOT = lambda: V_TB + 8       # noqa: E731  LDS address of this lane's broadcast word (TAB + 256 * wave + 4 * lane; readers take wave 0's lane 0)


def emit_grab_issue(E, tail):
    """Wave 0, lane 0: unit index <- atomic add on the home XCD's cursor of this list (device scope), into V_GRAB; every wave: V_SLOT.
    No wait here: the k-loop's counted vmcnt waits retire it (it is older than every load of the loop)."""
    skip = E.label("grabskip")
    # V_SLOT: every lane its own LDS word (TAB + 256 * wave + 4 * lane): the cursor's answer exists in wave 0's lane 0 only, and 64 lanes
    # storing different values to ONE address would leave some other lane's there
    E.e(f"v_mbcnt_lo_u32_b32 {vr(V_SLOT())}, -1, 0")
    E.e(f"v_mbcnt_hi_u32_b32 {vr(V_SLOT())}, -1, {vr(V_SLOT())}")
    E.e(f"s_lshl_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 8")
    E.e(f"s_add_u32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, {TAB}")
    E.e(f"v_lshl_add_u32 {vr(V_SLOT())}, {vr(V_SLOT())}, 2, {sr(S['TMP0'])}")
    E.e(f"s_cmp_lg_u32 {sr(S['WAVE'])}, 0")
    E.e(f"s_cbranch_scc1 {skip}")
    E.e(f"s_load_dwordx2 {sr(S['TMP64'], 2)}, {sr(S['KARG'], 2)}, {K_CURS}")
    E.e(f"s_lshl_b32 {sr(S['TMP0'])}, {sr(S['XHOME'])}, 2")
    if tail:
        E.e(f"s_add_u32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, 32")
    E.e(f"v_mov_b32 {vr(V_TB + 7)}, {sr(S['TMP0'])}")
    E.e(f"v_mov_b32 {vr(V_GRAB())}, 1")
    E.e("s_waitcnt lgkmcnt(0)")
    E.e("s_mov_b64 exec, 1")
    E.e(f"global_atomic_add {vr(V_GRAB())}, {vr(V_TB + 7)}, {vr(V_GRAB())}, {sr(S['TMP64'], 2)} sc0 sc1")
    E.e("s_mov_b64 exec, -1")
    E.e(f"{skip}:")


def emit_grab_read(E):
    """UNIT <- wave 0's broadcast word (written to LDS before the barrier the caller has passed)."""
    E.e(f"v_mov_b32 {vr(V_TB + 7)}, {TAB}")
    E.e(f"ds_read_b32 {vr(V_TB + 6)}, {vr(V_TB + 7)}")
    E.e("s_waitcnt lgkmcnt(0)")
    E.e(f"v_readfirstlane_b32 {sr(S['UNIT'])}, {vr(V_TB + 6)}")
    E.nops(2)


def emit_grab_sync(E, tail):
    """The first unit of a body: grab, publish, read.  The leading barrier keeps wave 0's write behind every earlier read of the word."""
    E.e("s_barrier")
    emit_grab_issue(E, tail)
    E.e("s_waitcnt vmcnt(0)")
    E.e(f"ds_write_b32 {vr(V_SLOT())}, {vr(V_GRAB())}")
    E.e("s_waitcnt lgkmcnt(0)")
    E.e("s_barrier")
    emit_grab_read(E)


def emit_make_record(E, tail):
    """Unit UNIT of the home XCD's list -> the 16-dword tile record in REC (REC+10 = 0: the list is exhausted), XD / WD / NK and the
    prefetch lanes.  What round 2's wrapper wrote into an LDS table per workgroup, now computed per unit from the static plan in the
    kernel arguments: which UNITS exist and what they compute is fixed by the shape (TailPlan in dit_gemm.hip); which workgroup takes
    which unit is decided by the cursor.  main list: whole tiles in raster order, then the k-range pieces of the left-over tiles;
    tail list: their 64-column pieces."""
    R = S["REC"]
    T0, T1, T2, T3, T4, T5, T6 = S["TMP0"], S["TMP1"], S["TMP2"], S["TMP64"], S["TMP64"] + 1, S["T"], S["KOFF"]
    P = S["XD"]                      # the plan (8 dwords) sits in XD / WD until they are set at the end
    none, have = E.label("recnone"), E.label("rechave")
    whole, gotlog = E.label("recwhole"), E.label("reclog")
    e = E.e
    e(f"s_load_dwordx8 {sr(R, 8)}, {sr(S['KARG'], 2)}, {K_DIMS1}")          # R0 nk, R1 tiles_m, R2 tiles_n, R3 gm, R4 gm*tiles_n, R5 magic, R6 nxcd, R7 grid
    e(f"s_load_dwordx4 {sr(R + 8, 4)}, {sr(S['KARG'], 2)}, {K_QD}")          # R8 qd, R9 rm, R10 per
    e("s_waitcnt lgkmcnt(0)")
    e(f"s_mov_b32 {sr(T0)}, {K_PLAN + 32}")
    e(f"s_cmp_lt_u32 {sr(S['XHOME'])}, {sr(R + 9)}")
    e(f"s_cselect_b32 {sr(T0)}, {K_PLAN}, {sr(T0)}")
    e(f"s_load_dwordx8 {sr(P, 8)}, {sr(S['KARG'], 2)}, {sr(T0)}")             # P0 n_whole, P1 n_main, P2 n_tail, P3 split, P4 split magic, P5 base, P6 extra, P7 full*per
    e(f"s_min_u32 {sr(T1)}, {sr(S['XHOME'])}, {sr(R + 9)}")
    e(f"s_mul_i32 {sr(T2)}, {sr(S['XHOME'])}, {sr(R + 8)}")
    e(f"s_add_u32 {sr(T2)}, {sr(T2)}, {sr(T1)}")                              # T2 = start: first logical tile of this XCD
    e("s_waitcnt lgkmcnt(0)")
    # in the list?  (and no workgroup takes more than 65535 units whatever the cursor says: an exit condition every wave reaches)
    e(f"s_cmp_lt_u32 {sr(S['UNIT'])}, {sr(P + (2 if tail else 1))}")
    e(f"s_cselect_b32 {sr(T1)}, 1, 0")
    e(f"s_cmp_lt_u32 {sr(S['NTILES'])}, 0xffff")
    e(f"s_cselect_b32 {sr(T1)}, {sr(T1)}, 0")
    e(f"s_cmp_eq_u32 {sr(T1)}, 1")
    e(f"s_cbranch_scc1 {have}")
    e(f"s_mov_b32 {sr(R + 10)}, 0")
    e(f"s_branch {none}")
    e(f"{have}:")
    # ---- unit -> logical tile T2, kind R10 (1 whole / column piece, 2 k-range piece), k range (T5 = k0, R0 = k-steps), T6 = piece / slot
    e(f"s_mov_b32 {sr(R + 10)}, 1")
    e(f"s_mov_b32 {sr(T5)}, 0")
    if tail:
        e(f"s_lshr_b32 {sr(T1)}, {sr(S['UNIT'])}, 2")
        e(f"s_and_b32 {sr(T6)}, {sr(S['UNIT'])}, 3")                            # T6 = column piece
        e(f"s_add_u32 {sr(T2)}, {sr(T2)}, {sr(P + 7)}")
        e(f"s_add_u32 {sr(T2)}, {sr(T2)}, {sr(T1)}")
    else:
        e(f"s_cmp_lt_u32 {sr(S['UNIT'])}, {sr(P)}")
        e(f"s_cbranch_scc1 {whole}")
        e(f"s_sub_u32 {sr(T6)}, {sr(S['UNIT'])}, {sr(P)}")                      # p: index among the k-range pieces
        e(f"s_mul_hi_u32 {sr(T1)}, {sr(T6)}, {sr(P + 4)}")                      # left-over tile index = p / split
        e(f"s_mul_i32 {sr(T0)}, {sr(T1)}, {sr(P + 3)}")
        e(f"s_sub_u32 {sr(T0)}, {sr(T6)}, {sr(T0)}")                            # j = p % split
        e(f"s_add_u32 {sr(T2)}, {sr(T2)}, {sr(P + 7)}")
        e(f"s_add_u32 {sr(T2)}, {sr(T2)}, {sr(T1)}")
        e(f"s_mov_b32 {sr(R + 10)}, 2")
        e(f"s_min_u32 {sr(T1)}, {sr(T0)}, {sr(P + 6)}")
        e(f"s_mul_i32 {sr(T5)}, {sr(T0)}, {sr(P + 5)}")
        e(f"s_add_u32 {sr(T5)}, {sr(T5)}, {sr(T1)}")                            # k0 = j * base + min(j, extra)
        e(f"s_cmp_lt_u32 {sr(T0)}, {sr(P + 6)}")
        e(f"s_cselect_b32 {sr(T1)}, 1, 0")
        e(f"s_add_u32 {sr(R)}, {sr(P + 5)}, {sr(T1)}")                          # k-steps = base + (j < extra)
        e(f"s_branch {gotlog}")
        e(f"{whole}:")
        e(f"s_add_u32 {sr(T2)}, {sr(T2)}, {sr(S['UNIT'])}")
        e(f"{gotlog}:")
    # ---- logical tile -> (mt, nt): grouped raster, gm row tiles at a time, m fastest inside a group
    e(f"s_mul_hi_u32 {sr(T1)}, {sr(T2)}, {sr(R + 5)}")                          # group
    e(f"s_mul_i32 {sr(T0)}, {sr(T1)}, {sr(R + 4)}")
    e(f"s_sub_u32 {sr(T2)}, {sr(T2)}, {sr(T0)}")                                # in_group
    e(f"s_mul_i32 {sr(T1)}, {sr(T1)}, {sr(R + 3)}")                             # group * gm = first row tile of the group
    e(f"s_sub_u32 {sr(T0)}, {sr(R + 1)}, {sr(T1)}")
    e(f"s_min_u32 {sr(T0)}, {sr(T0)}, {sr(R + 3)}")                             # row tiles in this group (1..4)
    e(f"s_mov_b32 {sr(T3)}, 0x40000000")
    e(f"s_cmp_eq_u32 {sr(T0)}, 2")
    e(f"s_cselect_b32 {sr(T3)}, 0x80000000, {sr(T3)}")
    e(f"s_cmp_eq_u32 {sr(T0)}, 3")
    e(f"s_cselect_b32 {sr(T3)}, 0x55555556, {sr(T3)}")
    e(f"s_mul_hi_u32 {sr(T3)}, {sr(T2)}, {sr(T3)}")
    e(f"s_cmp_eq_u32 {sr(T0)}, 1")
    e(f"s_cselect_b32 {sr(T3)}, {sr(T2)}, {sr(T3)}")                            # nt = in_group / rows
    e(f"s_mul_i32 {sr(T4)}, {sr(T3)}, {sr(T0)}")
    e(f"s_sub_u32 {sr(T2)}, {sr(T2)}, {sr(T4)}")
    e(f"s_add_u32 {sr(T1)}, {sr(T1)}, {sr(T2)}")                                # mt
    # prefetch lanes (REC+14), k range (REC+15), m0 (REC+11), n0 bytes (REC+12)
    e(f"s_and_b32 {sr(T0)}, {sr(T1)}, 3")
    e(f"s_lshl_b32 {sr(T0)}, {sr(T0)}, 8")
    if tail:
        e(f"s_or_b32 {sr(R + 14)}, {sr(T0)}, {sr(T6)}")
    else:
        e(f"s_and_b32 {sr(T2)}, {sr(T3)}, 7")
        e(f"s_or_b32 {sr(R + 14)}, {sr(T0)}, {sr(T2)}")
    e(f"s_lshl_b32 {sr(T0)}, {sr(R)}, 16")
    e(f"s_lshl_b32 {sr(T5)}, {sr(T5)}, 7")
    e(f"s_or_b32 {sr(R + 15)}, {sr(T0)}, {sr(T5)}")
    e(f"s_lshl_b32 {sr(R + 11)}, {sr(T1)}, 8")                                 # m0
    e(f"s_lshl_b32 {sr(T3)}, {sr(T3)}, 8")                                     # n0
    if tail:
        e(f"s_lshl_b32 {sr(T0)}, {sr(T6)}, 6")
        e(f"s_add_u32 {sr(T3)}, {sr(T3)}, {sr(T0)}")
    e(f"s_lshl_b32 {sr(R + 12)}, {sr(T3)}, 1")
    # ---- addresses.  T6 (k-range piece: p) -> workspace slot XHOME * per + p
    e(f"s_load_dwordx4 {sr(R, 4)}, {sr(S['KARG'], 2)}, {K_DIMS0}")              # R0 M, R1 lda bytes, R2 ldc bytes, R3 k bytes
    e(f"s_load_dwordx8 {sr(P, 8)}, {sr(S['KARG'], 2)}, {K_A}")                  # P0:1 a, P2:3 w, P4:5 c, P6:7 bias
    e(f"s_load_dwordx2 {sr(R + 4, 2)}, {sr(S['KARG'], 2)}, {K_WS}")
    e("s_waitcnt lgkmcnt(0)")
    TN = 64 if tail else 256
    e(f"s_sub_u32 {sr(T0)}, {sr(R)}, {sr(R + 11)}")
    e(f"s_min_u32 {sr(T0)}, {sr(T0)}, 256")
    e(f"s_sub_u32 {sr(T0)}, {sr(T0)}, 1")                                       # rows - 1
    e(f"s_mul_i32 {sr(R + 8)}, {sr(T0)}, {sr(R + 1)}")
    e(f"s_add_u32 {sr(R + 8)}, {sr(R + 8)}, {sr(R + 3)}")                       # x bytes
    e(f"s_mul_i32 {sr(R + 9)}, {sr(T0)}, {sr(R + 2)}")
    e(f"s_add_u32 {sr(R + 9)}, {sr(R + 9)}, {TN * 2}")                          # c bytes
    e(f"s_mul_i32 {sr(R + 13)}, {sr(R + 3)}, {TN}")                             # w bytes
    # bias + n0 bytes -> R6:7 (bias pointer P6:7)
    e(f"s_add_u32 {sr(R + 6)}, {sr(P + 6)}, {sr(R + 12)}")
    e(f"s_addc_u32 {sr(R + 7)}, {sr(P + 7)}, 0")
    # c + m0 * ldc bytes + n0 bytes -> T0:T1 (64-bit), or the piece's workspace slot
    e(f"s_mul_i32 {sr(T0)}, {sr(R + 11)}, {sr(R + 2)}")
    e(f"s_mul_hi_u32 {sr(T1)}, {sr(R + 11)}, {sr(R + 2)}")
    e(f"s_add_u32 {sr(T0)}, {sr(T0)}, {sr(R + 12)}")
    e(f"s_addc_u32 {sr(T1)}, {sr(T1)}, 0")
    e(f"s_add_u32 {sr(T0)}, {sr(T0)}, {sr(P + 4)}")
    e(f"s_addc_u32 {sr(T1)}, {sr(T1)}, {sr(P + 5)}")
    if not tail:
        notpiece = E.label("recnotpiece")
        e(f"s_cmp_lg_u32 {sr(R + 10)}, 2")
        e(f"s_cbranch_scc1 {notpiece}")
        e(f"s_load_dword {sr(T2)}, {sr(S['KARG'], 2)}, {K_QD + 8}")             # per
        e("s_waitcnt lgkmcnt(0)")
        e(f"s_mul_i32 {sr(T2)}, {sr(T2)}, {sr(S['XHOME'])}")
        e(f"s_add_u32 {sr(T2)}, {sr(T2)}, {sr(T6)}")                            # slot
        e(f"s_lshr_b32 {sr(T1)}, {sr(T2)}, 14")
        e(f"s_lshl_b32 {sr(T0)}, {sr(T2)}, 18")                                 # slot * PIECE_BYTES (2^18)
        e(f"s_add_u32 {sr(T0)}, {sr(T0)}, {sr(R + 4)}")
        e(f"s_addc_u32 {sr(T1)}, {sr(T1)}, {sr(R + 5)}")
        e(f"s_mov_b32 {sr(R + 9)}, {PIECE_BYTES}")
        e(f"{notpiece}:")
    e(f"s_mov_b32 {sr(R + 4)}, {sr(T0)}")
    e(f"s_mov_b32 {sr(R + 5)}, {sr(T1)}")
    # a + m0 * lda bytes -> XD ; w + n0 * k bytes -> WD   (P0:3 are XD0:3: read the pointers before they are overwritten)
    e(f"s_mul_i32 {sr(T0)}, {sr(R + 11)}, {sr(R + 1)}")
    e(f"s_mul_hi_u32 {sr(T1)}, {sr(R + 11)}, {sr(R + 1)}")
    e(f"s_mul_i32 {sr(T2)}, {sr(T3)}, {sr(R + 3)}")
    e(f"s_mul_hi_u32 {sr(T4)}, {sr(T3)}, {sr(R + 3)}")
    e(f"s_add_u32 {sr(T2)}, {sr(T2)}, {sr(P + 2)}")
    e(f"s_addc_u32 {sr(T4)}, {sr(T4)}, {sr(P + 3)}")
    e(f"s_add_u32 {sr(S['XD'])}, {sr(T0)}, {sr(P)}")
    e(f"s_addc_u32 {sr(S['XD'] + 1)}, {sr(T1)}, {sr(P + 1)}")
    e(f"s_mov_b32 {sr(S['XD'] + 2)}, {sr(R + 8)}")
    e(f"s_mov_b32 {sr(S['XD'] + 3)}, 0x00020000")
    e(f"s_mov_b32 {sr(S['WD'])}, {sr(T2)}")
    e(f"s_mov_b32 {sr(S['WD'] + 1)}, {sr(T4)}")
    e(f"s_mov_b32 {sr(S['WD'] + 2)}, {sr(R + 13)}")
    e(f"s_mov_b32 {sr(S['WD'] + 3)}, 0x00020000")
    # cooperative L2 prefetch lanes and the k range
    e(f"s_and_b32 {sr(PF_SX)}, {sr(R + 14)}, 0xff")
    e(f"s_lshr_b32 {sr(PF_SW)}, {sr(R + 14)}, 8")
    e(f"s_lshr_b32 {sr(S['NK'])}, {sr(R + 15)}, 16")
    e(f"s_add_u32 {sr(S['NTILES'])}, {sr(S['NTILES'])}, 1")
    e(f"{none}:")


def emit_load_record_b(E):
    """The record's output side -> the descriptors the epilogue of this tile uses (kept until that epilogue is through)."""
    E.e(f"s_mov_b64 {sr(S['CD'], 2)}, {sr(S['REC'] + 4, 2)}")
    E.e(f"s_mov_b32 {sr(S['CD'] + 2)}, {sr(S['REC'] + 9)}")
    E.e(f"s_mov_b64 {sr(S['BD'], 2)}, {sr(S['REC'] + 6, 2)}")
    E.e(f"s_mov_b32 {sr(S['M0ROW'])}, {sr(S['REC'] + 11)}")            # m0 of the tile (token class of a row: m0 + row < FIRST)
    E.e(f"s_mov_b32 {sr(S['KIND'])}, {sr(S['REC'] + 10)}")
    E.e(f"s_load_dwordx2 {sr(S['GD'], 2)}, {sr(S['KARG'], 2)}, {K_GATE}")
    E.e("s_waitcnt lgkmcnt(0)")
    E.e(f"s_add_u32 {sr(S['GD'])}, {sr(S['GD'])}, {sr(S['REC'] + 12)}")     # gate vector at the tile's first column
    E.e(f"s_addc_u32 {sr(S['GD'] + 1)}, {sr(S['GD'] + 1)}, 0")


def dma_piece(C, op, i, stage):
    base, dst, rs = (V_DW, S["WDST"], S["WD"]) if op == "W" else (V_DX, S["XDST"], S["XD"])
    src = vr(base + i) if C.ablate != "l1" else vr(255)          # l1 (timing only): every lane reads the first 16 bytes of the tile
    return [f"s_add_u32 m0, {sr(dst)}, {stage * C.stage_bytes + i * 1024}",
            "s_nop 0",
            f"buffer_load_dwordx4 {src}, {sr(rs, 4)}, {sr(S['KOFF'])} offen lds"]


def prefetch(C, op, slices_ahead):
    """One 128-byte line per lane: the k-slice `slices_ahead` beyond KOFF of this wave's rows, pulled into the XCD's L2 (the data
    is dropped).  The operand slices an XCD has not touched yet come from the Infinity Cache / HBM with a latency that the one
    k-step between an LDS-DMA and its barrier does not cover (measured: a k-loop that re-reads L2-resident slices runs 2 155
    cycles per step, the real one 2 430-2 600); the prefetch gives those misses pf_dist more steps."""
    base, rs = (V_PW, S["WD"]) if op == "W" else (V_PX, S["XD"])
    return f"buffer_load_dword {vr(V_PD)}, {vr(base)}, {sr(rs, 4)}, {sr(S['KOFF'])} offen offset:{128 * slices_ahead}"


def prefetch_lines(C, op, slices_ahead):
    if not C.pf_coop or (C.nb == 1 and op == "W"):      # a 64-column tail piece shares its weight rows with nobody
        return [prefetch(C, op, slices_ahead)]
    if C.nb == 1:      # ... and its activation rows with the 3 other pieces of its tile: every 4th row each
        pat, sh = "0x11111111", PF_SX
    else:
        pat, sh = ("0x01010101", PF_SX) if op == "X" else ("0x11111111", PF_SW)
    return [f"s_mov_b32 exec_lo, {pat}", f"s_mov_b32 exec_hi, {pat}", f"s_lshl_b64 exec, exec, {sr(sh)}",
            prefetch(C, op, slices_ahead), "s_mov_b64 exec, -1"]


def pf_ops(C):
    return ("X",) if C.pf_coop == 2 else ("W", "X")


def frag_read(C, op, blk, fset, stage, ks):
    """The ds_read_b128 lines that fetch fragment `blk` of operand `op` for k-substep `ks` (one line; two for the 32-byte e4m3 fragment)."""
    dst = fset + (0 if op == "W" else 4 * C.fr) + C.fr * blk
    if C.ring == 3:      # base set 0 = stage 0 (stage 1 through the immediate offset), set 1 = stage 2
        bset, extra = (1, 0) if stage == 2 else (0, stage * C.stage_bytes)
    else:
        bset, extra = stage, 0
    base = (V_WRD if op == "W" else V_XRD) + 4 * bset
    if FP8:
        return [f"ds_read_b128 {vr(dst + 4 * c, 4)}, {vr(base + 2 * ks + c)} offset:{blk * 4096 + extra}" for c in range(2)]
    return [f"ds_read_b128 {vr(dst, 4)}, {vr(base + ks)} offset:{blk * 4096 + extra}"]


def mfma_order(C, idx):
    """(ni, mi) of the idx-th MFMA of a k-substep.  serp: boustrophedon over mi, so that consecutive MFMAs always share one input
    operand (the W fragment inside a row of four, the X fragment at the turn)."""
    ni, j = idx >> 2, idx & 3
    return ni, (3 - j if (C.serp and (ni & 1)) else j)


def x_last_use(C, mi):
    """Position within a k-substep of the last MFMA that reads X fragment mi."""
    last_row = C.nb - 1
    return 4 * last_row + (3 - mi if (C.serp and (last_row & 1)) else mi)


def mfma(ni, mi, fset, zero_c=False):
    d = ar(acc(ni, mi), 16)
    if FP8:      # e4m3 x e4m3 (cbsz = blgp = 0), 64 k per instruction, 16 passes
        return f"v_mfma_f32_32x32x64_f8f6f4 {d}, {vr(fset + 8 * ni, 8)}, {vr(fset + 32 + 8 * mi, 8)}, {'0' if zero_c else d}"
    return f"v_mfma_f32_32x32x16_bf16 {d}, {vr(fset + 4 * ni, 4)}, {vr(fset + 16 + 4 * mi, 4)}, {'0' if zero_c else d}"


def build_iteration(E, C, stage, first, budget):
    """One K-step of 64: block 0 = last k-step of the previous stage (set F1), blocks 1..3 = k-steps 0..2 of this stage."""
    items = []
    add = items.append
    g16 = C.g16
    sets = [F1, F0, F1, F0][:C.nsub]
    for ks, (fset, busy_blk, need_blk) in enumerate([(F0, None, 1), (F1, 0, 2), (F0, 1, 3), (F1, 2, 4)][:C.nsub]):
        for op, nblk in (("W", C.nb), ("X", 4)):
            for blk in range(nblk):
                if busy_blk is None or (first and busy_blk == 0):
                    earliest = 0
                else:
                    earliest = g16 * busy_blk + (4 * blk + 3 if op == "W" else x_last_use(C, blk)) + 2
                need = g16 * need_blk + (4 * blk if op == "W" else blk)
                deadline = min(need - (2 if FP8 else 4), C.gaps - (2 if FP8 else 4))
                lines = frag_read(C, op, blk, fset, stage, ks)
                add(Item(f"rd{ks}{op}{blk}", lines, 2 * len(lines), earliest=earliest, deadline=max(deadline, earliest), lds=len(lines)))
    pieces = [("W", i) for i in range(C.npw)] + [("X", i) for i in range(8)]
    dma_step = min(C.dma_step, 0.5 * C.gaps / len(pieces))       # all pieces inside the first half of the step
    for n, (op, i) in enumerate(pieces if C.ablate != "nodma" else []):
        g0 = C.dma_first + int(n * dma_step)
        add(Item(f"dma{op}{i}", dma_piece(C, op, i, (stage + C.ring - 1) % C.ring), 12, earliest=g0, deadline=g0 + (4 if FP8 else 8)))
    n_pf = 0
    if C.pf_dist and C.ablate != "nodma":      # after every LDS-DMA piece in program order: the step's wait is vmcnt(n_pf)
        last = C.dma_first + int((len(pieces) - 1) * dma_step) + (4 if FP8 else 8)
        for n, op in enumerate(pf_ops(C)):
            add(Item(f"pf{op}", prefetch_lines(C, op, C.pf_dist), 8, earliest=min(last + 1 + 2 * n, C.gaps - 2),
                     deadline=min(last + (5 if FP8 else 9) + 2 * n, C.gaps - 1)))
            n_pf += 1
    # wave w stores V_GRAB to its LDS word every step (wave 0's is the next unit's index once the cursor's answer has landed — from the
    # second step on; the readers take it after the last step's barrier)
    add(Item("grabpub", [f"ds_write_b32 {vr(V_SLOT())}, {vr(V_GRAB())}"], 2, earliest=max(C.gaps - 8, 0), deadline=C.gaps - 2, lds=1))
    gaps, load = schedule(items, C.gaps, budget)
    lds_issued, lds_done, done_at = 0, 0, {}
    for g in range(C.gaps):
        b, idx = g // g16, g % g16
        ni, mi = mfma_order(C, idx)
        if not (first and b == 0):
            ks = b - 1
            if b >= 1:
                need = max(done_at[f"rd{ks}W{ni}"], done_at[f"rd{ks}X{mi}"])
                if need > lds_done:
                    E.e(f"s_waitcnt lgkmcnt({min(lds_issued - need, 15)})")
                    lds_done = need if lds_issued - need <= 15 else lds_issued - 15
            E.e(mfma(ni, mi, sets[b], zero_c=(first and b == 1)))
        for it in gaps[g]:
            for ln in it.lines:
                E.e(ln)
            lds_issued += it.lds
            if it.lds:
                done_at[it.name] = lds_issued
    if C.ablate != "samek":                        # samek (timing only): every k-step re-reads the tile's second slice (L2-resident)
        E.e(f"s_add_u32 {sr(S['KOFF'])}, {sr(S['KOFF'])}, 128")
    # ring of 2: everything but the prefetches; ring of 3: this step's pieces (slice t+2) may stay in flight, slice t+1 has landed
    n_keep = n_pf + (len(pieces) if C.ring == 3 and C.ablate != "nodma" else 0)
    E.e(f"s_waitcnt vmcnt({n_keep}) lgkmcnt(0)")
    E.e("s_barrier")
    return load


def emit_kloop(E, C, budget):
    """Tile's k-slice 0 is in stage 0 and visible (ring of 3: slice 1 is on its way to stage 1); KOFF = the next slice to fetch."""
    loop, done = E.label("kloop"), E.label("kdone")
    E.e(f"s_mov_b32 {sr(S['T'])}, 1")
    build_iteration(E, C, 0, True, budget)
    E.e(f"s_cmp_ge_u32 {sr(S['T'])}, {sr(S['NK'])}")
    E.e(f"s_cbranch_scc1 {done}")
    E.e(f"{loop}:")
    order = [(st + 1) % C.ring for st in range(C.ring)]           # stages 1, 0 / 1, 2, 0
    for stage in order:
        build_iteration(E, C, stage, False, budget)
        E.e(f"s_add_u32 {sr(S['T'])}, {sr(S['T'])}, 1")
        E.e(f"s_cmp_ge_u32 {sr(S['T'])}, {sr(S['NK'])}")
        E.e(f"s_cbranch_scc1 {done}" if stage != order[-1] else f"s_cbranch_scc0 {loop}")
    E.e(f"{done}:")
    for idx in range(C.g16):
        E.e(mfma(*mfma_order(C, idx), F1))
    if C.ring == 3:      # the last steps fetched past the end of K: those pieces must be down before the next tile's slices target the ring
        E.e("s_waitcnt vmcnt(0)")


def n_dma_k0(C):
    return (C.npw + 8) * (C.ring - 1) + len(pf_ops(C)) * C.pf_dist


def emit_dma_k0(E, C):
    """The tile's first k-slice -> stage 0 (ring of 3: and the second -> stage 1), and the L2 prefetch of the slices the first
    pf_dist steps will fetch."""
    E.e(f"s_and_b32 {sr(S['KOFF'])}, {sr(S['REC'] + 15)}, 0xffff")
    for sl in range(C.ring - 1):
        if sl:
            E.e(f"s_add_u32 {sr(S['KOFF'])}, {sr(S['KOFF'])}, 128")
        for op, n in (("W", C.npw), ("X", 8)):
            for i in range(n):
                for ln in dma_piece(C, op, i, sl):
                    E.e(ln)
    for d in range(1, C.pf_dist + 1):
        for op in pf_ops(C):
            for ln in prefetch_lines(C, op, d):
                E.e(ln)
    E.e(f"s_add_u32 {sr(S['KOFF'])}, {sr(S['KOFF'])}, 128")


def qoff(q):
    return 64 * (q >> 2) + 16 * (q & 3)


N_PREFETCH = lambda C: C.nb * 4 + 2 + (4 if FP8 else 0)          # noqa: E731  VMEM loads of emit_tile_prefetch
S_SCB, S_MLAST = S["ST1"], SB + 62           # fp8 form: base of the activation row scales (64-bit), M - 1  (the stamp counters' SGPRs: no --stamp there)


def emit_tile_prefetch(E, C):
    """Issued when a tile starts (BD / GD = the tile's bias / gate slices): the bias words in fragment layout -> V_BIASRAW, the
    lane's 8 gate values of both token classes in row-major layout -> V_G0 / V_G1 (GLD = 0 makes them equal; without a gate table
    GD points at the bias and the values are not used).  They land during the k-loop."""
    for q in range(C.nb * 4):
        E.e(f"buffer_load_dwordx2 {vr(V_BIASRAW + 2 * q, 2)}, {vr(V_COLB)}, {sr(S['BD'], 4)}, 0 offen offset:{qoff(q)}")
    E.e(f"buffer_load_dwordx4 {vr(V_G0, 4)}, {vr(V_GOFF)}, {sr(S['GD'], 4)}, 0 offen")
    E.e(f"buffer_load_dwordx4 {vr(V_G1, 4)}, {vr(V_GOFF)}, {sr(S['GD'], 4)}, {sr(S['GLD'])} offen")
    if FP8:      # scale_a of the lane's accumulator rows m0 + 128 (wave & 1) + 32 mi + r (clamped to the last row: those rows are never stored)
        E.e(f"v_mbcnt_lo_u32_b32 {vr(V_TB)}, -1, 0")
        E.e(f"v_mbcnt_hi_u32_b32 {vr(V_TB)}, -1, {vr(V_TB)}")
        E.e(f"v_and_b32 {vr(V_TB)}, 31, {vr(V_TB)}")
        E.e(f"s_and_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 1")
        E.e(f"s_lshl_b32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, 7")
        E.e(f"s_add_u32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, {sr(S['M0ROW'])}")
        E.e(f"v_add_u32 {vr(V_TB)}, {sr(S['TMP0'])}, {vr(V_TB)}")
        for mi in range(4):
            E.e(f"v_add_u32 {vr(V_TB + 1)}, {32 * mi}, {vr(V_TB)}")
            E.e(f"v_min_u32 {vr(V_TB + 1)}, {sr(S_MLAST)}, {vr(V_TB + 1)}")
            E.e(f"v_lshlrev_b32 {vr(V_TB + 2 + mi)}, 2, {vr(V_TB + 1)}")
        for mi in range(4):
            E.e(f"global_load_dword {vr(V_SC + mi)}, {vr(V_TB + 2 + mi)}, {sr(S_SCB, 2)}")


# GELU(tanh) in the row-major phase of the epilogue (FLAGS 4: ffn.0, models/wan_video_dit.py:208-209): y * sigmoid(2u),
# u = sqrt(2/pi) (y + 0.044715 y^3), as y * rcp(1 + exp2(y * (c1 + c2 y^2))) with c1 = -2 sqrt(2/pi) log2(e), c2 = 0.044715 c1: packed
# fp32 for the polynomial, v_exp_f32 / v_rcp_f32 (1 ulp) for the rest; y = -large gives y * rcp(inf) = -0, y = +large gives y * 1.
V_GC = V_XA            # v176..181: (c1, c1), (c2, c2), (1, 1)   [the residual buffers are idle in this mode]
V_GT = V_XA + 8        # v184..199: temporaries of two word pairs in flight
GELU_C1 = -2.0 * 0.7978845608028654 * 1.4426950408889634


def f32_bits(x):
    import struct
    return struct.unpack("<I", struct.pack("<f", x))[0]


def emit_gelu_consts(E):
    for i, val in enumerate((GELU_C1, GELU_C1, GELU_C1 * 0.044715, GELU_C1 * 0.044715, 1.0, 1.0)):
        E.e(f"v_mov_b32 {vr(V_GC + i)}, 0x{f32_bits(val):08x}")


def emit_gelu_words(E, words):
    """Four packed-bf16 words in place.  Two words (4 values) are advanced together so that a transcendental's result is first read
    two instructions later (trans -> VALU forwarding hazard on gfx940+)."""
    for w0, w1 in ((words[0], words[1]), (words[2], words[3])):
        xa, xb, ta, tb = V_GT, V_GT + 2, V_GT + 4, V_GT + 6
        for w, x in ((w0, xa), (w1, xb)):
            E.e(f"v_lshlrev_b32 {vr(x)}, 16, {vr(w)}")
            E.e(f"v_and_b32 {vr(x + 1)}, 0xffff0000, {vr(w)}")
        for x, t in ((xa, ta), (xb, tb)):
            E.e(f"v_pk_mul_f32 {vr(t, 2)}, {vr(x, 2)}, {vr(x, 2)}")
        for x, t in ((xa, ta), (xb, tb)):
            E.e(f"v_pk_fma_f32 {vr(t, 2)}, {vr(t, 2)}, {vr(V_GC + 2, 2)}, {vr(V_GC, 2)}")
        for x, t in ((xa, ta), (xb, tb)):
            E.e(f"v_pk_mul_f32 {vr(t, 2)}, {vr(t, 2)}, {vr(x, 2)}")
        for t in (ta, ta + 1, tb, tb + 1):
            E.e(f"v_exp_f32 {vr(t)}, {vr(t)}")
        E.e("s_nop 0")
        for t in (ta, tb):
            E.e(f"v_pk_add_f32 {vr(t, 2)}, {vr(t, 2)}, {vr(V_GC + 4, 2)}")
        for t in (ta, ta + 1, tb, tb + 1):
            E.e(f"v_rcp_f32 {vr(t)}, {vr(t)}")
        E.e("s_nop 0")
        for x, t in ((xa, ta), (xb, tb)):
            E.e(f"v_pk_mul_f32 {vr(x, 2)}, {vr(x, 2)}, {vr(t, 2)}")
        for w, x in ((w0, xa), (w1, xb)):
            E.e(f"v_cvt_pk_bf16_f32 {vr(w)}, {vr(x)}, {vr(x + 1)}")


def emit_epilogue(E, C, n_dma):
    """acc + bias -> bf16 (fragment layout) -> LDS -> row-major [-> x + gate * y | x + y] -> C with whole-row 16-byte stores.
    A lane of the 32x32 accumulator holds 4 consecutive columns of one row, so direct stores are 16-byte pieces of 32 different
    rows per instruction and the store ISSUE takes 20k cycles per tile (measured); through a per-wave LDS transpose (32-row blocks,
    row stride +16 B) every global access of the epilogue is a dwordx4 of 4 rows x 256 (192) contiguous bytes.
    n_dma LDS-DMA loads were issued right before; they are older than every VMEM access here, so the counted waits for residual
    words retire them too and the plain mode waits for them once at the end.  No wait on stores."""
    nq = C.nb * 4
    BIAS, TB, GS = 0, V_TB, V_TB + 6
    E.nops(8)
    for q in range(nq):
        lo, hi = V_BIASRAW + 2 * q, V_BIASRAW + 2 * q + 1
        E.e(f"v_and_b32 {vr(BIAS + 4 * q + 1)}, 0xffff0000, {vr(lo)}")
        E.e(f"v_lshlrev_b32 {vr(BIAS + 4 * q)}, 16, {vr(lo)}")
        E.e(f"v_and_b32 {vr(BIAS + 4 * q + 3)}, 0xffff0000, {vr(hi)}")
        E.e(f"v_lshlrev_b32 {vr(BIAS + 4 * q + 2)}, 16, {vr(hi)}")
    E.e(f"s_lshl_b32 {sr(S['TMP2'])}, {sr(S['LDC'])}, 2")                      # 4 rows
    E.e(f"s_mov_b32 {sr(S['TMP0'])}, 0")                                      # scalar row offset of the stores
    E.e(f"s_mov_b32 {sr(S['TMP1'])}, 0")                                      # ... of the residual loads

    def write_block(mi):
        for q in range(nq):
            ni, g = q >> 2, q & 3
            for j in range(4):
                E.e(f"v_accvgpr_read_b32 {vr(TB + j)}, {ar(acc(ni, mi, 4 * g + j))}")
            E.e("s_nop 0")
            if FP8:      # torch._scaled_mm: (acc * scale_a[row] * scale_b[col]) + bias, scale_b = 1: one fp32 rounding per operation
                sel = f"op_sel:[0,{mi & 1}] op_sel_hi:[1,{mi & 1}]"          # both halves take dword (mi & 1) of the scale pair
                E.e(f"v_pk_mul_f32 {vr(TB, 2)}, {vr(TB, 2)}, {vr(V_SC + (mi & 2), 2)} {sel}")
                E.e(f"v_pk_mul_f32 {vr(TB + 2, 2)}, {vr(TB + 2, 2)}, {vr(V_SC + (mi & 2), 2)} {sel}")
            E.e(f"v_pk_add_f32 {vr(TB, 2)}, {vr(TB, 2)}, {vr(BIAS + 4 * q, 2)}")
            E.e(f"v_pk_add_f32 {vr(TB + 2, 2)}, {vr(TB + 2, 2)}, {vr(BIAS + 4 * q + 2, 2)}")
            E.e(f"v_cvt_pk_bf16_f32 {vr(TB + 4)}, {vr(TB)}, {vr(TB + 1)}")
            E.e(f"v_cvt_pk_bf16_f32 {vr(TB + 5)}, {vr(TB + 2)}, {vr(TB + 3)}")
            E.e(f"ds_write_b64 {vr(V_LW)}, {vr(TB + 4, 2)} offset:{qoff(q)}")

    def read_block():
        for it in range(8):
            E.e(f"ds_read_b128 {vr(V_YR + 4 * it, 4)}, {vr(V_LR)} offset:{it * 4 * C.rs}")

    def load_x(buf):
        for it in range(8):
            E.e(f"buffer_load_dwordx4 {vr(buf + 4 * it, 4)}, {vr(V_SOFF)}, {sr(S['CD'], 4)}, {sr(S['TMP1'])} offen")
            E.e(f"s_add_u32 {sr(S['TMP1'])}, {sr(S['TMP1'])}, {sr(S['TMP2'])}")

    def store(it):
        E.e(f"buffer_store_dwordx4 {vr(V_YR + 4 * it, 4)}, {vr(V_SOFF)}, {sr(S['CD'], 4)}, {sr(S['TMP0'])} offen")
        E.e(f"s_add_u32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, {sr(S['TMP2'])}")

    plain, resid, done, raw, gelu = E.label("epiplain"), E.label("epiresid"), E.label("epidone"), E.label("epiraw"), E.label("epigelu")
    E.e(f"s_cmp_eq_u32 {sr(S['KIND'])}, 2")
    E.e(f"s_cbranch_scc1 {raw}")
    E.e(f"s_cmp_lt_u32 {sr(S['FLAGS'])}, 2")
    E.e(f"s_cbranch_scc1 {plain}")
    E.e(f"s_cmp_eq_u32 {sr(S['FLAGS'])}, 4")
    E.e(f"s_cbranch_scc1 {gelu}")
    E.e(f"s_cmp_eq_u32 {sr(S['FLAGS'])}, 3")
    E.e(f"s_cbranch_scc1 {resid}")
    for gated in (True, False):
        if not gated:
            E.e(f"{resid}:")
        else:      # class threshold of the row-major rows: row (of the tile, without 32mi + 4it) >= FIRST - m0 - 32mi - 4it -> class 1
            E.e(f"s_sub_i32 {sr(S['T'])}, {sr(S['FIRST'])}, {sr(S['M0ROW'])}")
        write_block(0)
        load_x(V_XA)
        for mi in range(4):
            xbuf = (V_XA, V_XB)[mi & 1]
            read_block()
            if mi < 3:
                write_block(mi + 1)
                load_x((V_XA, V_XB)[(mi + 1) & 1])
            E.e("s_waitcnt lgkmcnt(0)")
            E.e(f"s_waitcnt vmcnt({(8 if mi >= 1 else 0) + (8 if mi < 3 else 0)})")
            for it in range(8):
                if gated:
                    E.e(f"v_cmp_le_i32 vcc, {sr(S['T'])}, {vr(V_RROW)}")
                    E.e(f"s_sub_i32 {sr(S['T'])}, {sr(S['T'])}, 4")
                    E.e("s_nop 0")
                    for j in range(4):
                        E.e(f"v_cndmask_b32 {vr(GS + j)}, {vr(V_G0 + j)}, {vr(V_G1 + j)}, vcc")
                for j in range(4):
                    yw, xw = V_YR + 4 * it + j, xbuf + 4 * it + j
                    E.e(f"v_lshlrev_b32 {vr(TB)}, 16, {vr(yw)}")
                    E.e(f"v_and_b32 {vr(TB + 1)}, 0xffff0000, {vr(yw)}")
                    if gated:
                        E.e(f"v_lshlrev_b32 {vr(TB + 2)}, 16, {vr(GS + j)}")
                        E.e(f"v_and_b32 {vr(TB + 3)}, 0xffff0000, {vr(GS + j)}")
                        E.e(f"v_pk_mul_f32 {vr(TB, 2)}, {vr(TB, 2)}, {vr(TB + 2, 2)}")
                        E.e(f"v_cvt_pk_bf16_f32 {vr(TB + 4)}, {vr(TB)}, {vr(TB + 1)}")        # bf16(gate * y)
                        E.e(f"v_lshlrev_b32 {vr(TB)}, 16, {vr(TB + 4)}")
                        E.e(f"v_and_b32 {vr(TB + 1)}, 0xffff0000, {vr(TB + 4)}")
                    E.e(f"v_lshlrev_b32 {vr(TB + 2)}, 16, {vr(xw)}")
                    E.e(f"v_and_b32 {vr(TB + 3)}, 0xffff0000, {vr(xw)}")
                    E.e(f"v_pk_add_f32 {vr(TB, 2)}, {vr(TB, 2)}, {vr(TB + 2, 2)}")
                    E.e(f"v_cvt_pk_bf16_f32 {vr(yw)}, {vr(TB)}, {vr(TB + 1)}")
                store(it)
        E.e(f"s_branch {done}")
    # ---- plain: bias only; gelu: GELU(tanh) of the bf16-rounded y, rounded again (nn.Linear, then nn.GELU: two bf16 op boundaries)
    for act in (False, True):
        E.e(f"{gelu if act else plain}:")
        if act:
            emit_gelu_consts(E)
        write_block(0)
        for mi in range(4):
            read_block()
            if mi < 3:
                write_block(mi + 1)
            E.e("s_waitcnt lgkmcnt(0)")
            for it in range(8):
                if act:
                    emit_gelu_words(E, [V_YR + 4 * it + j for j in range(4)])
                store(it)
        E.e("s_waitcnt vmcnt(32)")                       # the LDS-DMA of the next tile's first k-slice has landed (no stall this late)
        E.e(f"s_branch {done}")
    # ---- a k-range piece: the fp32 accumulators as they stand -> the piece's workspace slot (CD), straight from the AGPRs.
    # Layout: dwordx4 number s = (ni*4 + mi)*4 + g of a wave at byte ((wave*(NB*16) + s)*64 + lane)*16: fg_gemm_reduce_kernel
    # (dit_gemm.hip) adds the pieces of a tile in k order and applies the epilogue.
    E.e(f"{raw}:")
    E.e(f"v_mbcnt_lo_u32_b32 {vr(TB)}, -1, 0")
    E.e(f"v_mbcnt_hi_u32_b32 {vr(TB)}, -1, {vr(TB)}")
    E.e(f"v_lshlrev_b32 {vr(TB)}, 4, {vr(TB)}")
    E.e(f"s_mul_i32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, {nq * 4 * 1024}")
    n = 0
    for ni in range(C.nb):
        for mi in range(4):
            for g in range(4):
                E.e(f"buffer_store_dwordx4 {ar(acc(ni, mi, 4 * g), 4)}, {vr(TB)}, {sr(S['CD'], 4)}, {sr(S['TMP0'])} offen")
                E.e(f"s_add_u32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, 1024")
                n += 1
                if n == 32:
                    E.e("s_waitcnt vmcnt(32)")          # the next tile's LDS-DMA (older than these stores) has landed
    if n < 32:
        E.e(f"s_waitcnt vmcnt({n})")
    E.e(f"{done}:")


def emit_program(E, C, stamp, budget, first, tail):
    """One body: takes units of its list (main: whole tiles + k-range pieces; tail: 64-column pieces) from the home XCD's cursor until
    the list is exhausted.  first: the body that starts the kernel; a later body re-derives every lane constant for its own tile width."""
    emit_setup(E, C, first)
    E.e(f"s_mov_b32 {sr(S['BD'] + 2)}, {C.nb * 2 * 32 * 2}")
    if FP8:
        assert not stamp, "the fp8 form keeps its scale base in the stamp counters' SGPRs"
    end = E.label("pend")
    tile_loop = E.label("ptile")
    nodma = E.label("pnodma")
    if stamp and first:
        for r in (ACC_K, ACC_E, ACC_W):
            E.e(f"s_mov_b32 {sr(r)}, 0")
        E.e(f"s_memrealtime {sr(S['TMP64'], 2)}")
        E.e("s_waitcnt lgkmcnt(0)")
        E.e(f"s_mov_b32 {sr(R_START)}, {sr(S['TMP64'])}")
        E.e(f"s_memtime {sr(S['TMP64'], 2)}")
        E.e("s_waitcnt lgkmcnt(0)")
        E.e(f"s_mov_b32 {sr(T_START)}, {sr(S['TMP64'])}")
        E.e(f"s_mov_b32 {sr(S['ST1'])}, {sr(S['TMP64'])}")
    if tail:      # most launches have no 64-column pieces at all: one scalar load instead of a cursor round trip
        E.e(f"s_load_dwordx4 {sr(S['REC'], 4)}, {sr(S['KARG'], 2)}, {K_QD}")
        E.e("s_waitcnt lgkmcnt(0)")
        E.e(f"s_mov_b32 {sr(S['TMP0'])}, {K_PLAN + 32 + 8}")
        E.e(f"s_cmp_lt_u32 {sr(S['XHOME'])}, {sr(S['REC'] + 1)}")
        E.e(f"s_cselect_b32 {sr(S['TMP0'])}, {K_PLAN + 8}, {sr(S['TMP0'])}")
        E.e(f"s_load_dword {sr(S['TMP0'])}, {sr(S['KARG'], 2)}, {sr(S['TMP0'])}")
        E.e("s_waitcnt lgkmcnt(0)")
        E.e(f"s_cmp_eq_u32 {sr(S['TMP0'])}, 0")
        E.e(f"s_cbranch_scc1 {end}")
    emit_grab_sync(E, tail)
    emit_make_record(E, tail)
    E.e(f"s_cmp_eq_u32 {sr(S['REC'] + 10)}, 0")
    E.e(f"s_cbranch_scc1 {end}")
    emit_dma_k0(E, C)
    emit_load_record_b(E)
    emit_tile_prefetch(E, C)
    E.e(f"s_waitcnt vmcnt({N_PREFETCH(C)})")        # the k-slice (and the L2 prefetches behind it) has landed; the bias / gate words may still be on their way
    E.e("s_barrier")
    E.e(f"{tile_loop}:")
    emit_grab_issue(E, tail)                          # the NEXT unit: its index travels while this tile's k-loop runs
    if stamp:
        emit_stamp(E, ACC_W)
    emit_kloop(E, C, budget)
    if stamp:
        emit_stamp(E, ACC_K)
    # next unit: its record (operand descriptors) and the LDS-DMA of its first k-slice go out BEFORE the epilogue of the current tile,
    # whose output / bias / gate descriptors (emit_load_record_b) stay those of the current tile until the epilogue is through
    emit_grab_read(E)
    emit_make_record(E, tail)
    E.e(f"s_cmp_eq_u32 {sr(S['REC'] + 10)}, 0")
    E.e(f"s_cbranch_scc1 {nodma}")
    emit_dma_k0(E, C)
    E.e(f"{nodma}:")
    emit_epilogue(E, C, n_dma_k0(C))
    if stamp:
        emit_stamp(E, ACC_E)
    E.e(f"s_cmp_eq_u32 {sr(S['REC'] + 10)}, 0")
    E.e(f"s_cbranch_scc1 {end}")
    emit_load_record_b(E)
    emit_tile_prefetch(E, C)
    E.e("s_barrier")                                 # every wave has waited for its LDS-DMA pieces inside the epilogue
    E.e(f"s_branch {tile_loop}")
    E.e(f"{end}:")
    E.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    E.e("s_barrier")                                 # nobody is still inside this body's LDS when the next one (or the stamp dump) starts


def emit_exit(E):
    """The last workgroup to get here (done counter, device scope) puts the cursors back to zero for the next launch on this stream."""
    skip = E.label("exitskip")
    E.e(f"s_cmp_lg_u32 {sr(S['WAVE'])}, 0")
    E.e(f"s_cbranch_scc1 {skip}")
    E.e(f"s_load_dwordx2 {sr(S['TMP64'], 2)}, {sr(S['KARG'], 2)}, {K_CURS}")
    E.e(f"s_load_dword {sr(S['TMP0'])}, {sr(S['KARG'], 2)}, {K_DIMS1 + 28}")       # grid
    E.e(f"v_mov_b32 {vr(0)}, 64")                                                  # byte offset of the done counter
    E.e(f"v_mov_b32 {vr(1)}, 1")
    E.e("s_waitcnt lgkmcnt(0)")
    E.e("s_mov_b64 exec, 1")
    E.e(f"global_atomic_add {vr(1)}, {vr(0)}, {vr(1)}, {sr(S['TMP64'], 2)} sc0 sc1")
    E.e("s_waitcnt vmcnt(0)")
    E.e(f"v_readfirstlane_b32 {sr(S['TMP1'])}, {vr(1)}")
    E.e("s_mov_b64 exec, -1")
    E.e(f"s_add_u32 {sr(S['TMP1'])}, {sr(S['TMP1'])}, 1")
    E.e(f"s_cmp_lg_u32 {sr(S['TMP1'])}, {sr(S['TMP0'])}")
    E.e(f"s_cbranch_scc1 {skip}")
    E.e(f"v_mbcnt_lo_u32_b32 {vr(0)}, -1, 0")                                      # lanes 0..16: one cursor word each (16 cursors + the counter)
    E.e(f"v_lshlrev_b32 {vr(0)}, 2, {vr(0)}")
    E.e(f"v_mov_b32 {vr(1)}, 0")
    E.e("s_mov_b64 exec, 0x1ffff")
    E.e(f"global_store_dword {vr(0)}, {vr(1)}, {sr(S['TMP64'], 2)} sc0 sc1")
    E.e("s_waitcnt vmcnt(0)")
    E.e("s_mov_b64 exec, -1")
    E.e(f"{skip}:")


def generate(nb, stamp, budget, tail=0):
    E = Emitter()
    emit_program(E, Cfg(nb), stamp, budget, True, False)
    if tail:      # the tiles of the last, partly filled round, cut into 256 x tail*64 pieces (the second list)
        emit_program(E, Cfg(tail), stamp, budget, False, True)
    emit_exit(E)
    if stamp:      # per wave: {kloop, epilogue, wait cycles, records passed, total cycles, total 100 MHz ticks} -> LDS table area (dead now)
        E.e(f"s_memtime {sr(S['TMP64'], 2)}")
        E.e("s_waitcnt lgkmcnt(0)")
        E.e(f"s_sub_u32 {sr(T_START)}, {sr(S['TMP64'])}, {sr(T_START)}")
        E.e(f"s_memrealtime {sr(S['TMP64'], 2)}")
        E.e("s_waitcnt lgkmcnt(0)")
        E.e(f"s_sub_u32 {sr(R_START)}, {sr(S['TMP64'])}, {sr(R_START)}")
        E.e("s_barrier")
        for n, r in enumerate((ACC_K, ACC_E, ACC_W, S["NTILES"], T_START, R_START)):
            E.e(f"v_mov_b32 {vr(8 + n)}, {sr(r)}")
        E.e(f"v_mov_b32 {vr(14)}, 0")
        E.e(f"v_mov_b32 {vr(15)}, 0")
        E.e(f"s_lshl_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 5")
        E.e(f"s_add_u32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, {TAB}")
        E.e(f"v_mov_b32 {vr(16)}, {sr(S['TMP0'])}")
        E.e(f"ds_write_b128 {vr(16)}, {vr(8, 4)}")
        E.e(f"ds_write_b128 {vr(16)}, {vr(12, 4)} offset:16")
        E.e("s_waitcnt lgkmcnt(0)")
    return E


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=4)
    ap.add_argument("--tail", type=int, default=0, help="second body for the table's second tile list: 256 x TAIL*64 tiles")
    ap.add_argument("--stamp", action="store_true")
    ap.add_argument("--budget", type=int, default=18)
    ap.add_argument("--dma-step", type=float, default=Cfg.dma_step)
    ap.add_argument("--dma-first", type=int, default=Cfg.dma_first)
    ap.add_argument("--ablate", default="")
    ap.add_argument("--pf-dist", type=int, default=Cfg.pf_dist)
    ap.add_argument("--pf-coop", type=int, default=Cfg.pf_coop)
    ap.add_argument("--serp", type=int, default=Cfg.serp)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"])
    a = ap.parse_args()
    if a.dtype == "fp8":
        configure_fp8()
    Cfg.dma_step, Cfg.dma_first, Cfg.ablate, Cfg.pf_dist, Cfg.pf_coop = a.dma_step, a.dma_first, a.ablate, a.pf_dist, a.pf_coop
    Cfg.serp = a.serp
    E = generate(a.nb, a.stamp, a.budget, a.tail)
    name = f"FG_GEMM_{'Q' if FP8 else 'P'}{a.nb}{a.tail or ''}"
    out = [f"// GENERATED by gen_gemm_p.py --nb {a.nb} --tail {a.tail} --dtype {a.dtype} : do not edit", f"#define {name}_ASM \\"]
    for ln in E.lines:
        out.append('    "%s\\n\\t" \\' % ln)
    out.append('    ""')
    # not used by the body and left to the compiler (it needs a VGPR for SGPR spills; the stamp build keeps values across the asm statement)
    free_v = (251,) if FP8 else (251, 252, 253, 254)
    regs = [f'"v{i}"' for i in range(256) if i not in free_v] + [f'"a{i}"' for i in range(256)] + \
           [f'"s{i}"' for i in range(SB, SB + NSREG + (4 if a.stamp else 1 if FP8 else 0))]
    out.append(f"#define {name}_CLOBBERS " + ", ".join(regs) + ', "vcc", "scc", "memory"')
    print("\n".join(out))


if __name__ == "__main__":
    main()
