#!/usr/bin/env python3
"""Generator of the hand-scheduled main body of attn_fwd_w4_kernel (csrc/attention.hip): gfx950 assembly, emitted as ONE inline
asm statement whose registers are all owned here (512-entry register file: 256 arch VGPRs + 256 AGPRs, one wave per SIMD).

    python3 gen_attn_w4.py --prescale 0|1 --name PREFIX > attn_w4[p]_asm.inc          (run by the Makefile, once per form)

Structure (the "4 waves x 64 query rows" form of /opt/skills/guides/cdna_hip_programming.md, Appendix B, built on this
repo's own LDS images and operand maps from attn_fwd_kernel<8,1>):
  * workgroup = 4 waves = 256 query rows of one head; each wave owns TWO 32-row q-blocks, so every K fragment and every V^T
    fragment read from LDS feeds two MFMAs (half the LDS bytes per FLOP of the 8 x 32 kernel);
  * AGPRs: O^T accumulators a[0:127], Q^T fragments a[128:191], the K fragments of one tile a[192:255];
    VGPRs: two score tiles (sA v[0:63], sB v[64:127]), packed P v[128:159], an 8-fragment V^T window v[160:191], ...
  * K / V tiles arrive by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, swizzle applied to the SOURCE
    address) into two 4-deep rings (K(t+4), V(t+3) issued in step t; one counted vmcnt + one barrier per tile);
  * per tile two phases of 32 MFMAs: S^T(t+1) = K(t+1) Q^T beside the exp2 / row sums / bf16 packs of tile t, then
    O^T += V(t)^T P(t)^T beside the row max of tile t+1 and the K(t+2) fragment reads; every other instruction is dealt to
    an MFMA gap by the list scheduler below (issue-cost budget per gap; at most one transcendental per gap);
  * deferred rescale (threshold 2^6) as an out-of-line rare path; ragged last tile masked by an out-of-line block.
--prescale 1: Q^T is multiplied by scale*log2(e) once and -max is the C operand of the first MFMA of every score chain, so the
exponent argument needs no VALU op (64 v_fma fewer per tile and wave).  The product is re-rounded to bf16, which is EXACT only when
scale*log2(e) is a power of two: the launcher (attention.hip) takes this form for such scales only — measured on peaked rows with a
general scale the second rounding costs 4x the error of the plain form (score error ~ |score| * 2^-9), see DESIGN.md §5 — and the
host arranges a power-of-two scale for self-attention by folding the remaining factor into q's RoPE table.
--prescale 0: exponent argument by v_fma on the fp32 scores of the exact bf16 operands, for every other scale.
"""
import argparse
import sys

THR = 6.0          # deferred rescale threshold (log2 units), same as kDeferLog2 of the HIP kernel
NVW = 8            # V^T fragment window (slots of 4 VGPRs)
DMA_COST, DMA_EVERY = 16, 3   # scheduler cost of one LDS-DMA piece (3 instructions) and the gap distance between pieces
GAP_BUDGET = 18    # issue-cost units a gap can hide besides its MFMA (guide: MFMA holds issue for 8 of its 32 cycles)
K_LDS, V_LDS, TILE = 0, 65536, 16384

# ---------------------------------------------------------------- register map
def sreg(buf, qi, sub, j):      # score tile register; buf 0 = sA, 1 = sB
    return buf * 64 + qi * 32 + sub * 16 + j
def preg(qi, kk, w):
    return 128 + qi * 16 + kk * 4 + w
def vwreg(slot, i):
    return 160 + slot * 4 + i
def negm(qi, j=0):
    return 192 + qi * 16 + j
V_KRD = 224          # 8 regs
V_VRD0, V_VRD1 = 232, 233
V_DK, V_DV = 234, 238            # 4 + 4 DMA source offsets
V_M = 242            # 2: running max (prescale: log2 units; else raw score units)
V_L = 244            # 4: l[qi][parity]
V_MT = 248           # 4: max temporaries [qi][sub]
V_THR, V_NINF, V_ROW, V_T0 = 252, 253, 254, 255
V_MB = 192           # (no prescale) m * scale_log2e per q-block: 2 regs in the unused NEGM area
def oacc(qi, db, j=0):
    return qi * 64 + db * 16 + j
def qfrag(qi, ks):
    return 128 + qi * 32 + ks * 4
def kfrag(sub, ks):
    return 192 + (sub * 8 + ks) * 4

SBASE = 40          # SGPRs SBASE .. SBASE+47 are owned by the asm body (clobbered); the compiler keeps its operands below
S = {k: v + SBASE for k, v in dict(
    QD=0, KD=4, VD=8, OD=12, MLD=16, WAVE=20, KSTR=21, VSTR=22, SCALE=23, T=24, TEND=25, TRAG=26, NVALID=27,
    LDQ=28, LDK=29, LDV=30, OROW=31, QROW0=32, PIECE=33, KDST=34, VDST=35, INVSCALE=36, TMP0=37, TMP1=38, TMP2=39,
    TMP64=40, SAVE64=42, NEXT=44, TENDM2=45, ARGS=48).items()}
NSREG = 60

def vr(lo, n=1):
    return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"
def ar(lo, n=1):
    return f"a{lo}" if n == 1 else f"a[{lo}:{lo + n - 1}]"
def sr(lo, n=1):
    return f"s{lo}" if n == 1 else f"s[{lo}:{lo + n - 1}]"


class Emitter:
    def __init__(self):
        self.lines = []
        self.uid = 0
    def e(self, text):
        self.lines.append(text)
    def label(self, stem):
        self.uid += 1
        return f"L_w4_{stem}_{self.uid}%="
    def nops(self, n):
        while n > 0:
            k = min(n, 8)
            self.e(f"s_nop {k - 1}")
            n -= k


# ---------------------------------------------------------------- list scheduler over MFMA gaps
class Item:
    def __init__(self, name, lines, cost, earliest=0, deadline=10 ** 9, deps=(), lds=0, trans=False):
        self.name, self.lines, self.cost = name, lines, cost
        self.earliest, self.deadline, self.deps = earliest, deadline, list(deps)      # deps: (Item, min gap distance)
        self.lds, self.trans = lds, trans          # lds: number of LDS return values this item issues (lgkmcnt tracking)
        self.gap = None

def schedule(items, ngaps, budget=None):
    budget = GAP_BUDGET if budget is None else budget
    load = [0] * ngaps
    ntrans = [0] * ngaps
    out = [[] for _ in range(ngaps)]
    order = sorted(range(len(items)), key=lambda i: (items[i].deadline, i))
    for i in order:
        it = items[i]
        g0 = it.earliest
        for dep, dist in it.deps:
            assert dep.gap is not None, f"{it.name}: dependency {dep.name} not scheduled yet"
            g0 = max(g0, dep.gap + dist)
        hi = min(it.deadline, ngaps - 1)
        assert g0 <= hi, f"{it.name}: window empty (earliest {g0} > deadline {hi})"
        pick = None
        for g in range(g0, hi + 1):
            if load[g] + it.cost <= budget and not (it.trans and ntrans[g] >= 1):
                pick = g
                break
        if pick is None:        # over budget everywhere in the window: least loaded gap (still obeying one transcendental per gap if possible)
            cands = [g for g in range(g0, hi + 1) if not (it.trans and ntrans[g] >= 1)] or list(range(g0, hi + 1))
            pick = min(cands, key=lambda g: (load[g], g))
        it.gap = pick
        load[pick] += it.cost
        ntrans[pick] += int(it.trans)
        out[pick].append((i, it))
    for g in range(ngaps):
        out[g].sort(key=lambda p: p[0])      # program order inside a gap = creation order (dependencies are created in order)
    return [[it for _, it in gap] for gap in out], load


# ---------------------------------------------------------------- pieces of the kernel
def emit_inputs(E):
    """Copy the asm operands (%0..%4 base pointers of q, k, v, o, ml; %5 wave, %6 t_begin, %7 t_end, %8 q_row0, %9 piece flag,
    %10 kernarg segment pointer) into fixed SGPRs, read the shape fields of AttnParams from the kernarg segment and build the
    buffer descriptors.  (The wrapper cannot pass everything as operands: the body owns most of the SGPR file.)"""
    for n, base in enumerate((S["QD"], S["KD"], S["VD"], S["OD"], S["MLD"])):
        E.e(f"s_mov_b64 {sr(base, 2)}, %{n}")
        E.e(f"s_mov_b32 {sr(base + 3)}, 0x00020000")
    for n, name in enumerate(["WAVE", "T", "TEND", "QROW0", "PIECE"]):
        E.e(f"s_mov_b32 {sr(S[name])}, %{5 + n}")
    A = S["ARGS"]           # 12 scratch SGPRs: ldq(2) ldk(2) | ldv(2) Nq(2) | Nkv(2) H scale
    E.e(f"s_load_dwordx4 {sr(A, 4)}, %10, 32")
    E.e(f"s_load_dwordx4 {sr(A + 4, 4)}, %10, 48")
    E.e(f"s_load_dwordx2 {sr(A + 8, 2)}, %10, 64")
    E.e(f"s_load_dword {sr(A + 10)}, %10, 72")
    E.e(f"s_load_dword {sr(S['SCALE'])}, %10, 96")
    E.e("s_waitcnt lgkmcnt(0)")
    E.e(f"s_lshl_b32 {sr(S['LDQ'])}, {sr(A)}, 1")
    E.e(f"s_lshl_b32 {sr(S['LDK'])}, {sr(A + 2)}, 1")
    E.e(f"s_lshl_b32 {sr(S['LDV'])}, {sr(A + 4)}, 1")
    E.e(f"s_lshl_b32 {sr(S['KSTR'])}, {sr(S['LDK'])}, 6")
    E.e(f"s_lshl_b32 {sr(S['VSTR'])}, {sr(S['LDV'])}, 6")
    E.e(f"s_and_b32 {sr(S['NVALID'])}, {sr(A + 8)}, 63")
    E.e(f"s_add_u32 {sr(S['TMP0'])}, {sr(A + 8)}, 63")
    E.e(f"s_lshr_b32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, 6")
    E.e(f"s_sub_u32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, 1")                  # nt - 1
    E.e(f"s_cmp_eq_u32 {sr(S['NVALID'])}, 0")
    E.e(f"s_cselect_b32 {sr(S['TRAG'])}, -1, {sr(S['TMP0'])}")
    E.e(f"s_lshl_b32 {sr(S['OROW'])}, {sr(A + 10)}, 8")                    # H * 128 * 2 bytes per output row
    E.e(f"s_sub_u32 {sr(S['TMP0'])}, {sr(A + 6)}, 1")                      # Nq - 1
    E.e(f"s_mul_i32 {sr(S['TMP1'])}, {sr(S['TMP0'])}, {sr(S['LDQ'])}")
    E.e(f"s_add_u32 {sr(S['QD'] + 2)}, {sr(S['TMP1'])}, 256")
    E.e(f"s_mul_i32 {sr(S['TMP1'])}, {sr(S['TMP0'])}, {sr(S['OROW'])}")
    E.e(f"s_add_u32 {sr(S['TMP1'])}, {sr(S['TMP1'])}, 256")                # direct: bytes of this head's output slice
    E.e(f"s_cmp_eq_u32 {sr(S['PIECE'])}, 0")
    E.e(f"s_cselect_b32 {sr(S['OD'] + 2)}, {sr(S['TMP1'])}, 0x20000")      # piece: 256 x 128 fp32
    E.e(f"s_cselect_b32 {sr(S['MLD'] + 2)}, 0, 0x800")                     # piece: 256 x 2 fp32
    E.e(f"s_sub_u32 {sr(S['TMP0'])}, {sr(A + 8)}, 1")                      # Nkv - 1
    E.e(f"s_mul_i32 {sr(S['TMP1'])}, {sr(S['TMP0'])}, {sr(S['LDK'])}")
    E.e(f"s_add_u32 {sr(S['KD'] + 2)}, {sr(S['TMP1'])}, 256")
    E.e(f"s_mul_i32 {sr(S['TMP1'])}, {sr(S['TMP0'])}, {sr(S['LDV'])}")
    E.e(f"s_add_u32 {sr(S['VD'] + 2)}, {sr(S['TMP1'])}, 256")
    E.e(f"v_rcp_f32 {vr(0)}, {sr(S['SCALE'])}")
    E.e("s_nop 1")
    E.e(f"v_readfirstlane_b32 {sr(S['INVSCALE'])}, {vr(0)}")
    E.e(f"s_lshl_b32 {sr(S['KDST'])}, {sr(S['WAVE'])}, 10")
    E.e(f"s_add_u32 {sr(S['VDST'])}, {sr(S['KDST'])}, {V_LDS}")
    E.nops(4)


def emit_lane_setup(E):
    """Per-lane constants: LDS read addresses, DMA source offsets, mask threshold."""
    L, R, HH, T0, T1, T2 = 0, 1, 2, 3, 4, 5        # temporaries in the (still unused) score area
    E.e(f"v_mbcnt_lo_u32_b32 {vr(L)}, -1, 0")
    E.e(f"v_mbcnt_hi_u32_b32 {vr(L)}, -1, {vr(L)}")
    E.e(f"v_and_b32 {vr(R)}, 31, {vr(L)}")
    E.e(f"v_lshrrev_b32 {vr(HH)}, 5, {vr(L)}")
    # k_rd[ks] = r*256 + (((2ks + hh) ^ (r & 15)) << 4)
    E.e(f"v_and_b32 {vr(T0)}, 15, {vr(R)}")
    E.e(f"v_lshlrev_b32 {vr(T1)}, 8, {vr(R)}")
    for ks in range(8):
        E.e(f"v_or_b32 {vr(T2)}, {2 * ks}, {vr(HH)}")
        E.e(f"v_xor_b32 {vr(T2)}, {vr(T2)}, {vr(T0)}")
        E.e(f"v_lshlrev_b32 {vr(T2)}, 4, {vr(T2)}")
        E.e(f"v_add_u32 {vr(V_KRD + ks)}, {vr(T1)}, {vr(T2)}")
    # v_rd0 = 64*row0 + 16*(ch ^ hh) + 8*(p&1) + V_LDS ; v_rd1 = 2048 + 64*row0 + 16*(ch ^ (2+hh)) + 8*(p&1) + V_LDS
    # row0 = 4hh + ((l&15)>>2), ch = 2*((l>>4)&1) + ((l&3)>>1), p = l&3
    E.e(f"v_bfe_u32 {vr(T0)}, {vr(L)}, 2, 2")                      # tr_q
    E.e(f"v_lshl_add_u32 {vr(T0)}, {vr(HH)}, 2, {vr(T0)}")          # row0
    E.e(f"v_bfe_u32 {vr(T1)}, {vr(L)}, 4, 1")                      # g1
    E.e(f"v_bfe_u32 {vr(T2)}, {vr(L)}, 1, 1")                      # p>>1
    E.e(f"v_lshl_add_u32 {vr(T1)}, {vr(T1)}, 1, {vr(T2)}")          # ch
    E.e(f"v_and_b32 {vr(T2)}, 1, {vr(L)}")                         # p&1
    E.e(f"v_lshlrev_b32 {vr(T2)}, 3, {vr(T2)}")
    E.e(f"v_lshl_add_u32 {vr(T2)}, {vr(T0)}, 6, {vr(T2)}")          # 64*row0 + 8*(p&1)
    E.e(f"v_xor_b32 {vr(T0)}, {vr(T1)}, {vr(HH)}")
    E.e(f"v_lshl_add_u32 {vr(V_VRD0)}, {vr(T0)}, 4, {vr(T2)}")
    E.e(f"v_add_u32 {vr(V_VRD0)}, {V_LDS}, {vr(V_VRD0)}")
    E.e(f"v_or_b32 {vr(T0)}, 2, {vr(HH)}")
    E.e(f"v_xor_b32 {vr(T0)}, {vr(T1)}, {vr(T0)}")
    E.e(f"v_lshl_add_u32 {vr(V_VRD1)}, {vr(T0)}, 4, {vr(T2)}")
    E.e(f"v_add_u32 {vr(V_VRD1)}, {V_LDS + 2048}, {vr(V_VRD1)}")
    # DMA K: row = 4w + (l>>4) (+16i), chunk = (l&15) ^ (4w + (l>>4)); vK[i] = row*ldk + chunk*16 + 16i*ldk + t_begin*kstride
    E.e(f"v_lshrrev_b32 {vr(T0)}, 4, {vr(L)}")
    E.e(f"s_lshl_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 2")
    E.e(f"v_add_u32 {vr(T0)}, {sr(S['TMP0'])}, {vr(T0)}")                # row
    E.e(f"v_and_b32 {vr(T1)}, 15, {vr(L)}")
    E.e(f"v_xor_b32 {vr(T1)}, {vr(T1)}, {vr(T0)}")                      # chunk
    E.e(f"v_lshlrev_b32 {vr(T1)}, 4, {vr(T1)}")
    E.e(f"v_mul_lo_u32 {vr(T0)}, {vr(T0)}, {sr(S['LDK'])}")
    E.e(f"v_add_u32 {vr(T0)}, {vr(T0)}, {vr(T1)}")
    E.e(f"s_sub_u32 {sr(S['TMP1'])}, {sr(S['T'])}, 1")                   # pre-increment form: one tile behind (wraps for t = 0)
    E.e(f"s_mul_i32 {sr(S['TMP1'])}, {sr(S['TMP1'])}, {sr(S['KSTR'])}")
    E.e(f"v_add_u32 {vr(V_DK)}, {sr(S['TMP1'])}, {vr(T0)}")
    E.e(f"s_lshl_b32 {sr(S['TMP2'])}, {sr(S['LDK'])}, 4")                # 16 rows
    for i in range(1, 4):
        E.e(f"v_add_u32 {vr(V_DK + i)}, {sr(S['TMP2'])}, {vr(V_DK + i - 1)}")
    # DMA V: row = 8(w>>1) + ((l>>2)&7) (+16i); chunk = 4(2(w&1) + (l>>5)) + ((l&3) ^ (2(w>>1) + ((l>>4)&1)))
    E.e(f"s_lshr_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 1")                 # w>>1
    E.e(f"s_and_b32 {sr(S['TMP1'])}, {sr(S['WAVE'])}, 1")                  # w&1
    E.e(f"v_bfe_u32 {vr(T0)}, {vr(L)}, 2, 3")
    E.e(f"s_lshl_b32 {sr(S['TMP2'])}, {sr(S['TMP0'])}, 3")
    E.e(f"v_add_u32 {vr(T0)}, {sr(S['TMP2'])}, {vr(T0)}")                # row
    E.e(f"v_bfe_u32 {vr(T1)}, {vr(L)}, 4, 1")
    E.e(f"s_lshl_b32 {sr(S['TMP2'])}, {sr(S['TMP0'])}, 1")
    E.e(f"v_add_u32 {vr(T1)}, {sr(S['TMP2'])}, {vr(T1)}")                # 2(w>>1) + ((l>>4)&1)
    E.e(f"v_and_b32 {vr(T2)}, 3, {vr(L)}")
    E.e(f"v_xor_b32 {vr(T1)}, {vr(T1)}, {vr(T2)}")                      # (l&3) ^ (...)
    E.e(f"s_lshl_b32 {sr(S['TMP2'])}, {sr(S['TMP1'])}, 1")
    E.e(f"v_add_u32 {vr(T2)}, {sr(S['TMP2'])}, {vr(HH)}")                # 2(w&1) + (l>>5)
    E.e(f"v_lshl_add_u32 {vr(T1)}, {vr(T2)}, 2, {vr(T1)}")              # chunk
    E.e(f"v_lshlrev_b32 {vr(T1)}, 4, {vr(T1)}")
    E.e(f"v_mul_lo_u32 {vr(T0)}, {vr(T0)}, {sr(S['LDV'])}")
    E.e(f"v_add_u32 {vr(T0)}, {vr(T0)}, {vr(T1)}")
    E.e(f"s_sub_u32 {sr(S['TMP1'])}, {sr(S['T'])}, 1")
    E.e(f"s_mul_i32 {sr(S['TMP1'])}, {sr(S['TMP1'])}, {sr(S['VSTR'])}")
    E.e(f"v_add_u32 {vr(V_DV)}, {sr(S['TMP1'])}, {vr(T0)}")
    E.e(f"s_lshl_b32 {sr(S['TMP2'])}, {sr(S['LDV'])}, 4")
    for i in range(1, 4):
        E.e(f"v_add_u32 {vr(V_DV + i)}, {sr(S['TMP2'])}, {vr(V_DV + i - 1)}")
    # mask threshold: key index constant c (= 32sub + (j&3) + 8(j>>2)) is valid iff c < nvalid - 4hh
    E.e(f"v_lshlrev_b32 {vr(T0)}, 2, {vr(HH)}")
    E.e(f"v_sub_u32 {vr(V_THR)}, {sr(S['NVALID'])}, {vr(T0)}")
    E.e(f"v_mov_b32 {vr(V_NINF)}, 0xff800000")
    # query row of this lane for q-block 0 (q-block 1: + 32 rows): q_row0 + wave*64 + r
    E.e(f"s_lshl_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 6")
    E.e(f"s_add_u32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, {sr(S['QROW0'])}")
    E.e(f"v_add_u32 {vr(V_ROW)}, {sr(S['TMP0'])}, {vr(R)}")
    # Q^T fragments: lane (r,hh) holds Q[row][16ks + 8hh + 0..7]: voffset = row*ldq + hh*16, imm ks*32
    E.e(f"v_mul_lo_u32 {vr(T0)}, {vr(V_ROW)}, {sr(S['LDQ'])}")
    E.e(f"v_lshl_add_u32 {vr(T0)}, {vr(HH)}, 4, {vr(T0)}")
    E.e(f"s_lshl_b32 {sr(S['TMP1'])}, {sr(S['LDQ'])}, 5")
    E.e(f"v_add_u32 {vr(T1)}, {sr(S['TMP1'])}, {vr(T0)}")
    E.nops(4)
    for qi, off in ((0, T0), (1, T1)):
        for ks in range(8):
            E.e(f"buffer_load_dwordx4 {ar(qfrag(qi, ks), 4)}, {vr(off)}, {sr(S['QD'], 4)}, 0 offen offset:{ks * 32}")
    # epilogue row offset: stash (row, hh) -> V_ROW keeps the row; hh is recomputed there


def dma_lines(kind, i, slot):
    """One LDS-DMA piece, pre-increment form: the source offset register holds the PREVIOUS tile's offset and is advanced by
    one tile right before its load; the advance also separates the M0 write from the load that reads M0 (one wait state)."""
    base, dst, rs, st = (V_DK, S["KDST"], S["KD"], S["KSTR"]) if kind == "K" else (V_DV, S["VDST"], S["VD"], S["VSTR"])
    return [f"s_add_u32 m0, {sr(dst)}, {slot * TILE + i * 4096}",
            f"v_add_u32 {vr(base + i)}, {sr(st)}, {vr(base + i)}",
            f"buffer_load_dwordx4 {vr(base + i)}, {sr(rs, 4)}, 0 offen lds"]


def emit_issue_tile(E, kind, slot):
    for i in range(4):
        for ln in dma_lines(kind, i, slot):
            E.e(ln)


def mfma_qk(buf, qi, sub, ks, first_c):
    d = vr(sreg(buf, qi, sub, 0), 16)
    return f"v_mfma_f32_32x32x16_bf16 {d}, {ar(kfrag(sub, ks), 4)}, {ar(qfrag(qi, ks), 4)}, {first_c if ks == 0 else d}"

def mfma_pv(qi, db, kk, slot):
    d = ar(oacc(qi, db), 16)
    return f"v_mfma_f32_32x32x16_bf16 {d}, {vr(vwreg(slot, 0), 4)}, {vr(preg(qi, kk, 0), 4)}, {d}"

def krd_line(sub, ks, slot):
    return f"ds_read_b128 {ar(kfrag(sub, ks), 4)}, {vr(V_KRD + ks)} offset:{K_LDS + slot * TILE + sub * 8192}"

def vrd_lines(f, slot_v, wslot):
    kk, db = f >> 2, f & 3
    off = slot_v * TILE + 4096 * kk + 512 * db
    return [f"ds_read_b64_tr_b16 {vr(vwreg(wslot, 0), 2)}, {vr(V_VRD0)} offset:{off}",
            f"ds_read_b64_tr_b16 {vr(vwreg(wslot, 2), 2)}, {vr(V_VRD1)} offset:{off}"]


def max_chain_lines(buf, qi, sub, with_prev=None):
    """8 ops reducing the 16 registers of one accumulator into V_MT[qi][sub] (the last one folds `with_prev` in)."""
    t = vr(V_MT + qi * 2 + sub)
    s = lambda j: vr(sreg(buf, qi, sub, j))      # noqa: E731
    lines = [f"v_max3_f32 {t}, {s(0)}, {s(1)}, {s(2)}"]
    for j in range(3, 15, 2):
        lines.append(f"v_max3_f32 {t}, {t}, {s(j)}, {s(j + 1)}")
    if with_prev is None:
        lines.append(f"v_max_f32 {t}, {t}, {s(15)}")
    else:
        lines.append(f"v_max3_f32 {t}, {t}, {s(15)}, {with_prev}")
    return lines

def pair_max_lines(reg, tmp):
    """reg <- max over the lane pair (l, l^32); 2 wait states between a VALU write and v_permlane32_swap."""
    return [f"v_mov_b32 {vr(tmp)}, {vr(reg)}", "s_nop 1", f"v_permlane32_swap_b32 {vr(tmp)}, {vr(reg)}", f"v_max_f32 {vr(reg)}, {vr(reg)}, {vr(tmp)}"]


def emit_rescale_block(E, buf_next, prescale, ret_label):
    """Rare path at the end of a step: some row's max of tile t+1 (V_MT[qi][1], already pair-reduced) outgrew the stale max by
    more than 2^THR.  Everything still at the old max is rescaled exactly once: O, l, and the scores of tile t+1."""
    E.nops(32)                                     # drain the matrix pipe before touching the O accumulators
    for qi in range(2):
        mt, m, t0 = V_MT + qi * 2 + 1, V_M + qi, V_T0
        if prescale:
            E.e(f"v_max_f32 {vr(t0)}, 0, {vr(mt)}")                     # delta = max(mt', 0)  (log2 units, relative to the stale max)
            E.e(f"v_add_f32 {vr(m)}, {vr(m)}, {vr(t0)}")
            for j in range(16):
                E.e(f"v_sub_f32 {vr(negm(qi, j))}, {vr(negm(qi, j))}, {vr(t0)}")
            for sub in range(2):
                for j in range(16):
                    r = sreg(buf_next, qi, sub, j)
                    E.e(f"v_sub_f32 {vr(r)}, {vr(r)}, {vr(t0)}")
            E.e(f"v_exp_f32 {vr(t0)}, -{vr(t0)}")                        # alpha
        else:
            E.e(f"v_max_f32 {vr(mt)}, {vr(m)}, {vr(mt)}")                # m_new (raw units)
            E.e(f"v_sub_f32 {vr(t0)}, {vr(m)}, {vr(mt)}")
            E.e(f"v_mul_f32 {vr(t0)}, {sr(S['SCALE'])}, {vr(t0)}")
            E.e(f"v_mov_b32 {vr(m)}, {vr(mt)}")
            E.e(f"v_mul_f32 {vr(V_MB + qi)}, {sr(S['SCALE'])}, {vr(m)}")
            E.e(f"v_exp_f32 {vr(t0)}, {vr(t0)}")
        E.e("s_nop 1")
        E.e(f"v_mul_f32 {vr(V_L + 2 * qi)}, {vr(V_L + 2 * qi)}, {vr(t0)}")
        E.e(f"v_mul_f32 {vr(V_L + 2 * qi + 1)}, {vr(V_L + 2 * qi + 1)}, {vr(t0)}")
        for j in range(64):
            a = oacc(qi, 0, j)
            E.e(f"v_accvgpr_read_b32 {vr(V_MT)}, {ar(a)}")
            E.e("s_nop 0")
            E.e(f"v_mul_f32 {vr(V_MT)}, {vr(V_MT)}, {vr(t0)}")
            E.e(f"v_accvgpr_write_b32 {ar(a)}, {vr(V_MT)}")
    E.nops(8)
    E.e(f"s_branch {ret_label}")


def emit_mask_block(E, buf_next, ret_label):
    """Out of line: tile t+1 is the ragged last tile: scores of keys >= Nkv become -inf (all 64 registers of both q-blocks)."""
    E.nops(24)                                     # the last S^T MFMAs must have written their results
    for sub in range(2):
        for j in range(16):
            c = 32 * sub + (j & 3) + 8 * (j >> 2)
            E.e(f"v_cmp_gt_i32 vcc, {vr(V_THR)}, {c}")
            for qi in range(2):
                r = sreg(buf_next, qi, sub, j)
                E.e(f"v_cndmask_b32 {vr(r)}, {vr(V_NINF)}, {vr(r)}, vcc")
    E.e(f"s_branch {ret_label}")


def build_step(E, phase, prescale, cold, flagged):
    """One pipeline step t with (t - t_begin) & 3 == phase.  sc = scores of tile t (buffer phase & 1), sn <- scores of tile t+1.
    flagged: the copy used for the last two steps of a range (tile t+1 may be the ragged last tile, or absent: S[NEXT]);
    the plain copy assumes an ordinary next tile.  `cold` collects out-of-line blocks (label, emitter function)."""
    cur, nxt = phase & 1, (phase & 1) ^ 1
    slot_v = phase                      # V(t) ring slot
    slot_k2 = (phase + 2) & 3           # K(t+2): fragment reads
    slot_kd = phase                     # K(t+4) DMA destination
    slot_vd = (phase + 3) & 3           # V(t+3) DMA destination
    items = []
    add = items.append

    # ---- LDS-DMA of K(t+4) and V(t+3): early in the step (longest flight before the vmcnt of the next step's end)
    for n, (kind, i, slot) in enumerate([("K", i, slot_kd) for i in range(4)] + [("V", i, slot_vd) for i in range(4)]):
        add(Item(f"dma{kind}{i}", dma_lines(kind, i, slot), DMA_COST, earliest=1 + DMA_EVERY * n, deadline=1 + DMA_EVERY * n + 6))
    # ---- softmax finish of tile t: element e -> (kk, qi, w): register sc[qi][kk>>1][8(kk&1) + w]
    exp_items = {}
    for kk in range(4):
        for qi in range(2):
            for w in range(8):
                r = sreg(cur, qi, kk >> 1, 8 * (kk & 1) + w)
                dl_pack = 32 + 8 * kk + qi - 2          # a whole MFMA between the pack and the MFMA that reads the word
                deps = []
                if not prescale:
                    fma = Item(f"fma{kk}{qi}{w}", [f"v_fma_f32 {vr(r)}, {vr(r)}, {sr(S['SCALE'])}, -{vr(V_MB + qi)}"], 4, deadline=dl_pack - 3)
                    add(fma)
                    deps = [(fma, 1)]
                ex = Item(f"exp{kk}{qi}{w}", [f"v_exp_f32 {vr(r)}, {vr(r)}"], 8, deadline=dl_pack - 2, deps=deps, trans=True)
                add(ex)
                exp_items[(kk, qi, w)] = ex
                lacc = V_L + 2 * qi + (w & 1)
                add(Item(f"add{kk}{qi}{w}", [f"v_add_f32 {vr(lacc)}, {vr(lacc)}, {vr(r)}"], 4, deadline=63, deps=[(ex, 1)]))
                if w & 1:
                    lo = sreg(cur, qi, kk >> 1, 8 * (kk & 1) + w - 1)
                    add(Item(f"pack{kk}{qi}{w >> 1}", [f"v_cvt_pk_bf16_f32 {vr(preg(qi, kk, w >> 1))}, {vr(lo)}, {vr(r)}"], 4,
                             deadline=dl_pack, deps=[(ex, 1), (exp_items[(kk, qi, w - 1)], 1)]))
    # ---- V^T fragment reads of tile t (window of NVW fragments); one counted wait per k-step group of 4 fragments
    for f in range(16):
        kk = f >> 2
        earliest = 0 if f < NVW else 32 + 2 * (f - NVW) + 2
        add(Item(f"vrd{f}", vrd_lines(f, slot_v, f % NVW), 2, earliest=earliest, deadline=max(earliest, 32 + 8 * kk - 4), lds=2))
    # ---- K fragment reads of tile t+2 into the AGPRs that tile t+1's phase A has finished with
    for i in range(16):
        sub, ks = i >> 3, i & 7
        add(Item(f"krd{i}", [krd_line(sub, ks, slot_k2)], 2, earliest=2 * i + 3, deadline=58, lds=1))
    # ---- row max of tile t+1 (phase B: its scores are complete, and a ragged tile has been masked, by then)
    for qi in range(2):
        c0 = max_chain_lines(nxt, qi, 0)
        c1 = max_chain_lines(nxt, qi, 1, with_prev=vr(V_MT + qi * 2))
        prev = None
        for n, ln in enumerate(c0 + c1):
            it = Item(f"max{qi}_{n}", [ln], 4, earliest=34, deadline=58, deps=[(prev, 0)] if prev else [])
            add(it)
            prev = it
        add(Item(f"pmax{qi}", pair_max_lines(V_MT + qi * 2 + 1, V_T0), 14, earliest=36, deadline=61, deps=[(prev, 1)]))
    gaps, load = schedule(items, 64)

    # ---- emission, with lgkmcnt tracking (LDS results return in order)
    lds_issued = 0                # LDS results requested so far in this step (the step starts with none outstanding)
    lds_done = 0                  # ... of which this many are known complete (by the last counted wait)
    vrd_done_at = {}              # fragment -> value of lds_issued right after its second read
    lab_mask, lab_mask_ret = E.label("mask"), E.label("maskret")
    lab_resc, lab_resc_ret = E.label("rescale"), E.label("rescret")
    lab_skip = E.label("nodecide")
    first_c = [vr(negm(qi), 16) if prescale else "0" for qi in range(2)]
    for g in range(64):
        if g == 32 and flagged:
            # between the phases: tile t+1 ragged? (scalar flag computed at the start of the step)
            E.e(f"s_cmp_eq_u32 {sr(S['NEXT'])}, 2")
            E.e(f"s_cbranch_scc1 {lab_mask}")
            E.e(f"{lab_mask_ret}:")
        if g < 32:
            sub, ks, qi = g >> 4, (g >> 1) & 7, g & 1
            E.e(mfma_qk(nxt, qi, sub, ks, first_c[qi]))
        else:
            p = g - 32
            kk, db, qi = p >> 3, (p >> 1) & 3, p & 1
            f = kk * 4 + db
            need = max(vrd_done_at[ff] for ff in range(4 * kk, 4 * kk + 4)) if (db == 0 and qi == 0) else vrd_done_at[f]
            if need > lds_done:
                E.e(f"s_waitcnt lgkmcnt({min(lds_issued - need, 15)})")
                lds_done = max(need, lds_issued - 15) if lds_issued - need > 15 else need
            E.e(mfma_pv(qi, db, kk, f % NVW))
        for it in gaps[g]:
            for ln in it.lines:
                E.e(ln)
            lds_issued += it.lds
            if it.name.startswith("vrd"):
                vrd_done_at[int(it.name[3:])] = lds_issued
    # ---- end of step: rescale decision (flagged copy: skipped when there is no next tile), waits, barrier
    if flagged:
        E.e(f"s_cmp_eq_u32 {sr(S['NEXT'])}, 0")
        E.e(f"s_cbranch_scc1 {lab_skip}")
    if prescale:
        E.e(f"v_max_f32 {vr(V_T0)}, {vr(V_MT + 1)}, {vr(V_MT + 3)}")
    else:
        E.e(f"v_sub_f32 {vr(V_T0)}, {vr(V_MT + 1)}, {vr(V_M)}")
        E.e(f"v_sub_f32 {vr(V_MT)}, {vr(V_MT + 3)}, {vr(V_M + 1)}")
        E.e(f"v_max_f32 {vr(V_T0)}, {vr(V_T0)}, {vr(V_MT)}")
        E.e(f"v_mul_f32 {vr(V_T0)}, {sr(S['SCALE'])}, {vr(V_T0)}")
    E.e(f"v_cmp_lt_f32 vcc, {THR}, {vr(V_T0)}")
    E.e("s_waitcnt vmcnt(8) lgkmcnt(0)")
    E.e(f"s_cbranch_vccnz {lab_resc}")
    E.e(f"{lab_resc_ret}:")
    if flagged:
        E.e(f"{lab_skip}:")
        E.e("s_waitcnt vmcnt(8) lgkmcnt(0)")
    E.e("s_barrier")
    if flagged:
        cold.append((lab_mask, lambda EE, b=nxt, r=lab_mask_ret: emit_mask_block(EE, b, r)))
    cold.append((lab_resc, lambda EE, b=nxt, r=lab_resc_ret: emit_rescale_block(EE, b, prescale, r)))
    return load


def emit_next_flags(E):
    """S[NEXT] for step t: 0 = no tile t+1 in this range, 2 = tile t+1 is the ragged last tile, 1 = ordinary."""
    E.e(f"s_add_u32 {sr(S['TMP0'])}, {sr(S['T'])}, 1")
    E.e(f"s_mov_b32 {sr(S['NEXT'])}, 1")
    E.e(f"s_cmp_eq_u32 {sr(S['TMP0'])}, {sr(S['TRAG'])}")
    E.e(f"s_cselect_b32 {sr(S['NEXT'])}, 2, {sr(S['NEXT'])}")
    E.e(f"s_cmp_ge_u32 {sr(S['TMP0'])}, {sr(S['TEND'])}")
    E.e(f"s_cselect_b32 {sr(S['NEXT'])}, 0, {sr(S['NEXT'])}")


def emit_prologue(E, prescale):
    if STAMP:
        E.e(f"s_memtime {sr(SBASE + 46, 2)}")          # whole-workgroup stamp (SBASE+46..47 are otherwise unused)
    emit_inputs(E)
    emit_lane_setup(E)
    # ring fill: K(tb .. tb+3) -> slots 0..3, V(tb .. tb+2) -> slots 0..2   (28 pieces per wave).  Issue order = need order:
    # K(tb), V(tb), K(tb+1), K(tb+2) must be visible before the first step; K(tb+3), V(tb+1), V(tb+2) are first read in the
    # second step and are covered by the first step's end-of-step vmcnt(8) + barrier
    emit_issue_tile(E, "K", 0)
    emit_issue_tile(E, "V", 0)
    emit_issue_tile(E, "K", 1)
    emit_issue_tile(E, "K", 2)
    emit_issue_tile(E, "K", 3)
    for slot in (1, 2):
        emit_issue_tile(E, "V", slot)
    # accumulators
    for a in range(128):
        E.e(f"v_accvgpr_write_b32 {ar(a)}, 0")
    for i in range(4):
        E.e(f"v_mov_b32 {vr(V_L + i)}, 0")
    # Q fragments have landed (16 loads issued before the 28 DMA pieces)
    E.e("s_waitcnt vmcnt(28)")
    if prescale:
        for a in range(128, 192):
            E.e(f"v_accvgpr_read_b32 {vr(0)}, {ar(a)}")
            E.e("s_nop 0")
            E.e(f"v_lshlrev_b32 {vr(1)}, 16, {vr(0)}")
            E.e(f"v_and_b32 {vr(2)}, 0xffff0000, {vr(0)}")
            E.e(f"v_mul_f32 {vr(1)}, {sr(S['SCALE'])}, {vr(1)}")
            E.e(f"v_mul_f32 {vr(2)}, {sr(S['SCALE'])}, {vr(2)}")
            E.e(f"v_cvt_pk_bf16_f32 {vr(0)}, {vr(1)}, {vr(2)}")
            E.e(f"v_accvgpr_write_b32 {ar(a)}, {vr(0)}")
    # K(tb) landed (first 4 of the 28 pieces) -> visible to every wave -> fragments -> S^T(tb)
    E.e("s_waitcnt vmcnt(24)")
    E.e("s_barrier")
    for i in range(16):
        E.e(krd_line(i >> 3, i & 7, 0))
    E.e("s_waitcnt lgkmcnt(0)")
    E.nops(4)
    for g in range(32):
        sub, ks, qi = g >> 4, (g >> 1) & 7, g & 1
        E.e(mfma_qk(0, qi, sub, ks, "0"))
    # K(tb+1) fragments for phase A of the first step; K(tb), V(tb), K(tb+1), K(tb+2) landed and visible (12 pieces still fly)
    E.e("s_waitcnt vmcnt(12)")
    E.e("s_barrier")
    E.nops(16)                                          # the MFMAs above have read their K fragments
    for i in range(16):
        E.e(krd_line(i >> 3, i & 7, 1))
    # row max of tile tb -> running max; scores made relative to it (prescale) / mb (plain)
    E.nops(16)
    for qi in range(2):
        for ln in max_chain_lines(0, qi, 0) + max_chain_lines(0, qi, 1, with_prev=vr(V_MT + qi * 2)):
            E.e(ln)
        for ln in pair_max_lines(V_MT + qi * 2 + 1, V_T0):
            E.e(ln)
        E.e(f"v_mov_b32 {vr(V_M + qi)}, {vr(V_MT + qi * 2 + 1)}")
        if prescale:
            for j in range(16):
                E.e(f"v_sub_f32 {vr(negm(qi, j))}, 0, {vr(V_M + qi)}")
            for sub in range(2):
                for j in range(16):
                    r = sreg(0, qi, sub, j)
                    E.e(f"v_sub_f32 {vr(r)}, {vr(r)}, {vr(V_M + qi)}")
        else:
            E.e(f"v_mul_f32 {vr(V_MB + qi)}, {sr(S['SCALE'])}, {vr(V_M + qi)}")
    E.e("s_waitcnt lgkmcnt(0)")
    E.nops(4)


def emit_epilogue(E, prescale):
    E.nops(32)                                         # last O^T MFMAs retired
    lab_piece, lab_done = E.label("piece"), E.label("done")
    HH, T1 = 0, 1
    E.e(f"v_mbcnt_lo_u32_b32 {vr(HH)}, -1, 0")
    E.e(f"v_mbcnt_hi_u32_b32 {vr(HH)}, -1, {vr(HH)}")
    E.e(f"v_lshrrev_b32 {vr(HH)}, 5, {vr(HH)}")
    # l over the lane pair
    for qi in range(2):
        E.e(f"v_add_f32 {vr(V_L + 2 * qi)}, {vr(V_L + 2 * qi)}, {vr(V_L + 2 * qi + 1)}")
        E.e(f"v_mov_b32 {vr(V_T0)}, {vr(V_L + 2 * qi)}")
        E.e("s_nop 1")
        E.e(f"v_permlane32_swap_b32 {vr(V_T0)}, {vr(V_L + 2 * qi)}")
        E.e(f"v_add_f32 {vr(V_L + 2 * qi)}, {vr(V_L + 2 * qi)}, {vr(V_T0)}")
    E.e(f"s_cmp_lg_u32 {sr(S['PIECE'])}, 0")
    E.e(f"s_cbranch_scc1 {lab_piece}")
    # ---- direct: normalise, bf16, store 8 bytes per (db, g); lane holds O[row][32db + 8g + 4hh + 0..3]
    for qi in range(2):
        E.e(f"v_rcp_f32 {vr(V_T0)}, {vr(V_L + 2 * qi)}")
        E.e(f"v_add_u32 {vr(T1)}, {32 * qi}, {vr(V_ROW)}")
        E.e(f"v_mul_lo_u32 {vr(T1)}, {vr(T1)}, {sr(S['OROW'])}")
        E.e(f"v_lshl_add_u32 {vr(T1)}, {vr(HH)}, 3, {vr(T1)}")
        for db in range(4):
            for g in range(4):
                base = 8 + (db * 4 + g) % 8 * 6          # rotating temporaries v8..v55
                for j in range(4):
                    E.e(f"v_accvgpr_read_b32 {vr(base + j)}, {ar(oacc(qi, db, 4 * g + j))}")
                E.e("s_nop 0")
                for j in range(4):
                    E.e(f"v_mul_f32 {vr(base + j)}, {vr(base + j)}, {vr(V_T0)}")
                E.e(f"v_cvt_pk_bf16_f32 {vr(base + 4)}, {vr(base)}, {vr(base + 1)}")
                E.e(f"v_cvt_pk_bf16_f32 {vr(base + 5)}, {vr(base + 2)}, {vr(base + 3)}")
                E.e(f"buffer_store_dwordx2 {vr(base + 4, 2)}, {vr(T1)}, {sr(S['OD'], 4)}, 0 offen offset:{64 * db + 16 * g}")
                if (db * 4 + g) % 8 == 7:
                    E.e("s_waitcnt vmcnt(0)")
    E.e(f"s_branch {lab_done}")
    # ---- piece: un-normalised fp32 partial (row-local index) + (max in raw score units, row sum)
    E.e(f"{lab_piece}:")
    E.e(f"s_lshl_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 6")
    E.e(f"s_add_u32 {sr(S['TMP0'])}, {sr(S['TMP0'])}, {sr(S['QROW0'])}")
    E.e(f"v_subrev_u32 {vr(2)}, {sr(S['TMP0'])}, {vr(V_ROW)}")           # r
    E.e(f"s_lshl_b32 {sr(S['TMP0'])}, {sr(S['WAVE'])}, 6")
    E.e(f"v_add_u32 {vr(2)}, {sr(S['TMP0'])}, {vr(2)}")                  # row-local = wave*64 + r
    for qi in range(2):
        E.e(f"v_add_u32 {vr(T1)}, {32 * qi}, {vr(2)}")
        E.e(f"v_lshlrev_b32 {vr(3)}, 3, {vr(T1)}")                       # ml offset: row*8
        E.e(f"v_lshlrev_b32 {vr(T1)}, 9, {vr(T1)}")                      # row*512
        E.e(f"v_lshl_add_u32 {vr(T1)}, {vr(HH)}, 4, {vr(T1)}")
        for db in range(4):
            for g in range(4):
                base = 8 + (db * 4 + g) % 8 * 4
                for j in range(4):
                    E.e(f"v_accvgpr_read_b32 {vr(base + j)}, {ar(oacc(qi, db, 4 * g + j))}")
                E.e("s_nop 1")
                E.e(f"buffer_store_dwordx4 {vr(base, 4)}, {vr(T1)}, {sr(S['OD'], 4)}, 0 offen offset:{128 * db + 32 * g}")
                if (db * 4 + g) % 8 == 7:
                    E.e("s_waitcnt vmcnt(0)")
        if prescale:
            E.e(f"v_mul_f32 {vr(4)}, {sr(S['INVSCALE'])}, {vr(V_M + qi)}")
        else:
            E.e(f"v_mov_b32 {vr(4)}, {vr(V_M + qi)}")
        E.e(f"v_mov_b32 {vr(5)}, {vr(V_L + 2 * qi)}")
        E.e(f"s_mov_b64 {sr(S['SAVE64'], 2)}, exec")
        E.e("s_mov_b64 exec, 0xffffffff")                                 # lanes 0..31 (hh == 0)
        E.e(f"buffer_store_dwordx2 {vr(4, 2)}, {vr(3)}, {sr(S['MLD'], 4)}, 0 offen")
        E.e("s_nop 1")
        E.e(f"s_mov_b64 exec, {sr(S['SAVE64'], 2)}")
    E.e(f"{lab_done}:")
    E.e("s_waitcnt vmcnt(0) lgkmcnt(0)")


STAMP = False       # diagnostic build: every workgroup overwrites the first 16 bytes of its first output row with
                    # (shader cycles of the tile loop, 100 MHz ticks of the tile loop, steps, 0): tools/attn_ab.py --stamp
ABLATE = set()      # timing-only experiments (WRONG results), tools/build_attn_variant.sh: dma, barrier, exp, lds, max, valu


def ablate_line(ln):
    """Main-loop line -> replacement under the active ablations (None = drop)."""
    op = ln.split()[0]
    if "dma" in ABLATE and (op == "buffer_load_dwordx4" or ln.startswith("s_add_u32 m0")):
        return None
    if "barrier" in ABLATE and op == "s_barrier":
        return None
    if "vmcnt" in ABLATE and ln.startswith("s_waitcnt vmcnt(8)"):
        return "s_waitcnt lgkmcnt(0)"
    if "exp" in ABLATE and op == "v_exp_f32":
        return None
    if "lds" in ABLATE and op in ("ds_read_b128", "ds_read_b64_tr_b16"):
        return None
    if "max" in ABLATE and op in ("v_max3_f32", "v_max_f32", "v_permlane32_swap_b32"):
        return None
    if "valu" in ABLATE and op in ("v_fma_f32", "v_add_f32", "v_cvt_pk_bf16_f32", "v_exp_f32", "v_max3_f32", "v_max_f32"):
        return None
    return ln


def generate(prescale):
    E = Emitter()
    cold = []
    emit_prologue(E, prescale)
    bulk, done = E.label("bulk"), E.label("done_steps")
    tails = [E.label(f"tail{p}") for p in range(4)]
    # plain steps while t < t_end - 2 (their next tile exists and is not the ragged one), then the flagged copy for the last two
    E.e(f"s_sub_u32 {sr(S['TENDM2'])}, {sr(S['TEND'])}, 2")
    if STAMP:
        A = S["ARGS"]
        E.e(f"s_mov_b32 {sr(A + 8)}, {sr(S['T'])}")
        E.e(f"s_memtime {sr(A, 2)}")
        E.e(f"s_memrealtime {sr(A + 2, 2)}")
        E.e("s_waitcnt lgkmcnt(0)")
    E.e(f"s_cmp_ge_i32 {sr(S['T'])}, {sr(S['TENDM2'])}")
    E.e(f"s_cbranch_scc1 {tails[0]}")
    E.e(f"{bulk}:")
    loop_start = len(E.lines)
    loads = []
    for phase in range(4):
        loads.append(build_step(E, phase, prescale, cold, flagged=False))
        E.e(f"s_add_u32 {sr(S['T'])}, {sr(S['T'])}, 1")
        E.e(f"s_cmp_ge_i32 {sr(S['T'])}, {sr(S['TENDM2'])}")
        if phase < 3:
            E.e(f"s_cbranch_scc1 {tails[phase + 1]}")
        else:
            E.e(f"s_cbranch_scc0 {bulk}")
    for phase in range(4):
        E.e(f"{tails[phase]}:")
        emit_next_flags(E)
        build_step(E, phase, prescale, cold, flagged=True)
        E.e(f"s_add_u32 {sr(S['T'])}, {sr(S['T'])}, 1")
        E.e(f"s_cmp_ge_u32 {sr(S['T'])}, {sr(S['TEND'])}")
        E.e(f"s_cbranch_scc1 {done}")
        if phase == 3:
            E.e(f"s_branch {tails[0]}")
    if ABLATE:
        body = [ablate_line(ln) for ln in E.lines[loop_start:]]
        E.lines[loop_start:] = [ln for ln in body if ln is not None]
    E.e(f"{done}:")
    if STAMP:
        A = S["ARGS"]
        E.e(f"s_memtime {sr(A + 4, 2)}")
        E.e(f"s_memrealtime {sr(A + 6, 2)}")
        E.e("s_waitcnt lgkmcnt(0)")
        E.e(f"s_sub_u32 {sr(A)}, {sr(A + 4)}, {sr(A)}")
        E.e(f"s_sub_u32 {sr(A + 2)}, {sr(A + 6)}, {sr(A + 2)}")
        E.e(f"s_sub_u32 {sr(A + 8)}, {sr(S['TEND'])}, {sr(A + 8)}")
    emit_epilogue(E, prescale)
    if STAMP:
        A = S["ARGS"]
        skip = E.label("nostamp")
        E.e(f"s_cmp_lg_u32 {sr(S['WAVE'])}, 0")
        E.e(f"s_cbranch_scc1 {skip}")
        E.e(f"s_cmp_lg_u32 {sr(S['PIECE'])}, 0")
        E.e(f"s_cbranch_scc1 {skip}")
        E.e(f"s_mul_i32 {sr(S['TMP0'])}, {sr(S['QROW0'])}, {sr(S['OROW'])}")
        E.e(f"v_mov_b32 {vr(8)}, {sr(A)}")
        E.e(f"v_mov_b32 {vr(9)}, {sr(A + 2)}")
        E.e(f"v_mov_b32 {vr(10)}, {sr(A + 8)}")
        E.e(f"s_memtime {sr(S['TMP64'], 2)}")
        E.e("s_waitcnt lgkmcnt(0)")
        E.e(f"s_sub_u32 {sr(S['TMP1'])}, {sr(S['TMP64'])}, {sr(SBASE + 46)}")
        E.e(f"v_mov_b32 {vr(11)}, {sr(S['TMP1'])}")
        E.e(f"v_mov_b32 {vr(12)}, {sr(S['TMP0'])}")
        E.e("s_mov_b64 exec, 1")
        E.e(f"buffer_store_dwordx4 {vr(8, 4)}, {vr(12)}, {sr(S['OD'], 4)}, 0 offen")
        E.e("s_waitcnt vmcnt(0)")
        E.e("s_mov_b64 exec, -1")
        E.e(f"{skip}:")
    end = E.label("end")
    E.e(f"s_branch {end}")
    for lab, fn in cold:
        E.e(f"{lab}:")
        fn(E)
    E.e(f"{end}:")
    return E, loads


def main():
    global GAP_BUDGET, DMA_EVERY, STAMP
    ap = argparse.ArgumentParser()
    ap.add_argument("--prescale", type=int, default=1)
    ap.add_argument("--name", default="FG_ATTN_W4", help="prefix of the generated macros (<name>_ASM, <name>_CLOBBERS)")
    ap.add_argument("--ablate", default="", help="comma list of timing-only ablations (wrong results): dma,barrier,vmcnt,exp,lds,max,valu")
    ap.add_argument("--stamp", action="store_true", help="diagnostic build: cycle / clock stamps of the tile loop into the output (corrupts it)")
    ap.add_argument("--budget", type=int, default=GAP_BUDGET)
    ap.add_argument("--dma-every", type=int, default=DMA_EVERY)
    ap.add_argument("--report", action="store_true", help="print the per-gap issue-cost load of the four steps to stderr")
    a = ap.parse_args()
    GAP_BUDGET = a.budget
    DMA_EVERY = a.dma_every
    STAMP = a.stamp
    ABLATE.update(x for x in a.ablate.split(",") if x)
    E, loads = generate(bool(a.prescale))
    if a.report:
        for p, ld in enumerate(loads):
            print(f"step phase {p}: gap loads {ld} (max {max(ld)}, sum {sum(ld)})", file=sys.stderr)
    out = ["// GENERATED by gen_attn_w4.py --prescale %d --name %s : do not edit" % (a.prescale, a.name),
           "#define %s_ASM \\" % a.name]
    for ln in E.lines:
        out.append('    "%s\\n\\t" \\' % ln)
    out.append('    ""')
    regs = [f'"v{i}"' for i in range(256)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(SBASE, SBASE + NSREG)]
    out.append(f"#define {a.name}_CLOBBERS " + ", ".join(regs) + ', "vcc", "scc", "memory"')
    print("\n".join(out))


if __name__ == "__main__":
    main()
