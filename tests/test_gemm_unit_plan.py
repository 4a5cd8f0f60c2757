"""CPU check of the persistent GEMM's unit scheduler: the scalar program gen_gemm_p.py emits to turn (home XCD, list, unit index) into a
tile record is interpreted here instruction by instruction against a fake kernel-argument block, for every unit of every XCD of the DiT
shapes and the ragged test shapes, and compared with the plan written out in Python (TailPlan / the raster of dit_gemm.hip).  Guards the
address arithmetic (a wrong record is an out-of-bounds access on the GPU) without a GPU."""
import os
import re
import struct
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fairygen_amd", "csrc"))
import gen_gemm_p as G  # noqa: E402

M32 = 0xFFFFFFFF


def tail_plan(total, xcd, nxcd, per, nk, have_ws, col_cut):
    qd, rm = divmod(total, nxcd)
    start = xcd * qd + min(xcd, rm)
    ln = qd + (1 if xcd < rm else 0)
    full = ln // per
    rem = ln - full * per
    split = 1
    if have_ws and rem > 0:
        split = max(1, min(per // rem, nk // 2))
    cut = col_cut if (split == 1 and col_cut > 1 and rem > 0 and rem * col_cut <= per) else 1
    return dict(start=start, len=ln, full=full, rem=rem, split=split, cut=cut)


def build_sched(M, N, k_bytes, lda_bytes, nxcd=8, per=32, have_ws=True, gm=4):
    """The GemmSched block as launch_gemm fills it (pointers are fake, far apart, so that every field of a record is checkable)."""
    tiles_m, tiles_n = (M + 255) // 256, N // 256
    total = tiles_m * tiles_n
    nk = k_bytes // 128
    ws = have_ws and nk >= 96
    gmtn = gm * tiles_n
    s = dict(a=0x1000_0000_0000, w=0x2000_0000_0000, c=0x3000_0000_0000, bias=0x4000_0000_0000, ws=0x5000_0000_0000, cursors=0, gate=0, scale=0,
             M=M, lda_bytes=lda_bytes, ldc_bytes=N * 2, k_bytes=k_bytes, nk=nk, tiles_m=tiles_m, tiles_n=tiles_n, gm=gm, gmtn=gmtn,
             gmtn_magic=((1 << 32) + gmtn - 1) // gmtn, nxcd=nxcd, grid=nxcd * per, qd=total // nxcd, rm=total % nxcd, per=per, flags=0,
             first_rows=0, gate_ld_bytes=0, gate_bytes=0, m_last=M - 1, total=total, have_ws=ws)
    plans = []
    for which in range(2):
        xcd = 0 if which == 0 else s["rm"]
        tp = tail_plan(total, xcd, nxcd, per, nk, ws, 4)
        fp = tp["full"] * per
        n_whole = fp + (tp["rem"] if tp["split"] == 1 and tp["cut"] == 1 else 0)
        plans.append(dict(n_whole=n_whole, n_main=n_whole + (tp["rem"] * tp["split"] if tp["split"] > 1 else 0),
                          n_tail=tp["rem"] * 4 if tp["cut"] > 1 else 0, split=tp["split"],
                          split_magic=((1 << 32) + tp["split"] - 1) // tp["split"] if tp["split"] > 1 else 0,
                          base=nk // tp["split"], extra=nk % tp["split"], full_per=fp))
    s["plans"] = plans
    blob = struct.pack("<8Q", s["a"], s["w"], s["c"], s["bias"], s["ws"], s["cursors"], s["gate"], s["scale"])
    blob += struct.pack("<16I", M, lda_bytes, N * 2, k_bytes, nk, tiles_m, tiles_n, gm, gmtn, s["gmtn_magic"], nxcd, nxcd * per,
                        s["qd"], s["rm"], per, 0)
    blob += struct.pack("<4I", 0, 0, 0, M - 1)
    for pl in plans:
        blob += struct.pack("<8I", pl["n_whole"], pl["n_main"], pl["n_tail"], pl["split"], pl["split_magic"], pl["base"], pl["extra"], pl["full_per"])
    return s, blob


def expected_record(s, xcd, tail, unit):
    """Python restatement of the unit lists (dit_gemm.hip TailPlan + raster); None when the list is exhausted."""
    tp = tail_plan(s["total"], xcd, s["nxcd"], s["per"], s["nk"], s["have_ws"], 4)
    pl = s["plans"][0 if xcd < s["rm"] else 1]
    kind, k0, nk, piece, slot = 1, 0, s["nk"], 0, None
    if tail:
        if unit >= pl["n_tail"]:
            return None
        logical, piece = tp["start"] + tp["full"] * s["per"] + unit // 4, unit % 4
    elif unit >= pl["n_main"]:
        return None
    elif unit < pl["n_whole"]:
        logical = tp["start"] + unit
    else:
        p = unit - pl["n_whole"]
        ti, j = divmod(p, tp["split"])
        logical, kind, slot = tp["start"] + tp["full"] * s["per"] + ti, 2, xcd * s["per"] + p
        base, extra = divmod(s["nk"], tp["split"])
        k0, nk = j * base + min(j, extra), base + (1 if j < extra else 0)
    gm, tn = s["gm"], s["tiles_n"]
    group, in_group = divmod(logical, gm * tn)
    rows_in_group = min(gm, s["tiles_m"] - group * gm)
    mt, nt = group * gm + in_group % rows_in_group, in_group // rows_in_group
    assert 0 <= mt < s["tiles_m"] and 0 <= nt < tn, (logical, mt, nt)
    TN = 64 if tail else 256
    m0, n0 = mt * 256, nt * 256 + piece * 64
    rows = min(256, s["M"] - m0)
    cp = s["c"] + m0 * s["ldc_bytes"] + n0 * 2 if kind == 1 else s["ws"] + slot * 262144
    return dict(xp=s["a"] + m0 * s["lda_bytes"], wp=s["w"] + n0 * s["k_bytes"], cp=cp, bp=s["bias"] + n0 * 2,
                x_bytes=(rows - 1) * s["lda_bytes"] + s["k_bytes"], c_bytes=(rows - 1) * s["ldc_bytes"] + TN * 2 if kind == 1 else 262144,
                kind=kind, m0=m0, n0_bytes=n0 * 2, w_bytes=TN * s["k_bytes"], lanes=(piece if tail else nt & 7) | ((mt & 3) << 8),
                krange=(k0 * 128) | (nk << 16), logical=logical)


class Scalar:
    """Interpreter of the SALU / SMEM subset emit_make_record uses."""

    def __init__(self, lines, kernarg):
        self.lines, self.mem = lines, kernarg
        self.labels = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}

    def run(self, regs):
        s, scc, pc, steps = dict(regs), 0, 0, 0

        def val(tok):
            tok = tok.strip()
            if re.fullmatch(r"s\d+", tok):
                return s.get(int(tok[1:]), 0xDEADBEEF)
            return int(tok, 0) & M32

        def rng(tok):
            m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok.strip())
            return int(m.group(1)), int(m.group(2)) - int(m.group(1)) + 1

        while pc < len(self.lines):
            ln = self.lines[pc]
            pc += 1
            steps += 1
            assert steps < 10000
            if ln.endswith(":") or ln.startswith("s_waitcnt") or ln.startswith("s_nop"):
                continue
            op, rest = ln.split(None, 1)
            a = [t.strip() for t in rest.split(",")]
            if op in ("s_load_dword", "s_load_dwordx2", "s_load_dwordx4", "s_load_dwordx8"):
                n = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8}[op]
                dst = int(a[0][1:]) if n == 1 else rng(a[0])[0]
                base, _ = rng(a[1])
                assert s[base] == 0 and s[base + 1] == 0, "kernarg pointer"
                off = val(a[2])
                for i in range(n):
                    s[dst + i] = struct.unpack_from("<I", self.mem, off + 4 * i)[0]
            elif op == "s_mov_b32":
                s[int(a[0][1:])] = val(a[1])
            elif op in ("s_add_u32", "s_addc_u32"):
                r = val(a[1]) + val(a[2]) + (scc if op == "s_addc_u32" else 0)
                s[int(a[0][1:])], scc = r & M32, int(r > M32)
            elif op == "s_sub_u32":
                r = val(a[1]) - val(a[2])
                s[int(a[0][1:])], scc = r & M32, int(r < 0)
            elif op == "s_mul_i32":
                s[int(a[0][1:])] = (val(a[1]) * val(a[2])) & M32
            elif op == "s_mul_hi_u32":
                s[int(a[0][1:])] = (val(a[1]) * val(a[2])) >> 32
            elif op == "s_min_u32":
                x, y = val(a[1]), val(a[2])
                s[int(a[0][1:])], scc = min(x, y), int(x < y)
            elif op in ("s_lshl_b32", "s_lshr_b32", "s_and_b32", "s_or_b32"):
                x, y = val(a[1]), val(a[2])
                r = {"s_lshl_b32": (x << (y & 31)) & M32, "s_lshr_b32": x >> (y & 31), "s_and_b32": x & y, "s_or_b32": x | y}[op]
                s[int(a[0][1:])], scc = r, int(r != 0)
            elif op.startswith("s_cmp_"):
                x, y = val(a[0]), val(a[1])
                scc = int({"s_cmp_lt_u32": x < y, "s_cmp_eq_u32": x == y, "s_cmp_lg_u32": x != y, "s_cmp_ge_u32": x >= y}[op])
            elif op == "s_cselect_b32":
                s[int(a[0][1:])] = val(a[1]) if scc else val(a[2])
            elif op == "s_branch":
                pc = self.labels[a[0]]
            elif op in ("s_cbranch_scc1", "s_cbranch_scc0"):
                if scc == (1 if op.endswith("1") else 0):
                    pc = self.labels[a[0]]
            else:
                raise AssertionError(f"instruction outside the interpreted subset: {ln}")
        return s


def program(tail):
    E = G.Emitter()
    G.emit_make_record(E, tail)
    return [ln.replace("%=", "") for ln in E.lines]


@pytest.mark.parametrize("M,N,k_bytes,nxcd,per", [
    (27280, 3072, 6144, 8, 32), (27280, 9216, 6144, 8, 32), (27280, 14336, 6144, 8, 32), (27280, 3072, 28672, 8, 32),     # the DiT shapes, bf16
    (3410, 3072, 6144, 8, 32), (3410, 3072, 28672, 8, 32), (6820, 14336, 3072, 8, 32),                                      # token shards (the last: e4m3)
    (700, 768, 512, 8, 32), (4200, 4096, 1024, 8, 32), (600, 3072, 28672, 8, 32), (66200, 256, 12288, 8, 32), (10500, 2048, 256, 8, 32),
    (27280, 3072, 6144, 8, 38), (513, 256, 256, 1, 40)])                                                                   # other CU counts / one XCD
def test_scalar_record_program_matches_the_plan(M, N, k_bytes, nxcd, per):
    lda = k_bytes + 128
    s, blob = build_sched(M, N, k_bytes, lda, nxcd=nxcd, per=per)
    R, XD, WD = G.S["REC"], G.S["XD"], G.S["WD"]
    seen = set()
    for tail in (False, True):
        prog = Scalar(program(tail), blob)
        for xcd in range(nxcd):
            unit = 0
            while True:
                want = expected_record(s, xcd, tail, unit)
                out = prog.run({G.S["KARG"]: 0, G.S["KARG"] + 1: 0, G.S["XHOME"]: xcd, G.S["UNIT"]: unit, G.S["NTILES"]: 0})
                if want is None:
                    assert out[R + 10] == 0, (xcd, tail, unit)
                    break
                rec = [out[R + i] for i in range(16)]
                got = dict(xp=out[XD] | out[XD + 1] << 32, wp=out[WD] | out[WD + 1] << 32, cp=rec[4] | rec[5] << 32, bp=rec[6] | rec[7] << 32,
                           x_bytes=rec[8], c_bytes=rec[9], kind=rec[10], m0=rec[11], n0_bytes=rec[12], w_bytes=rec[13], lanes=rec[14], krange=rec[15])
                for key, v in got.items():
                    assert v == want[key], (xcd, tail, unit, key, hex(v), hex(want[key]))
                assert out[XD + 2] == want["x_bytes"] and out[WD + 2] == want["w_bytes"] and out[G.S["NK"]] == want["krange"] >> 16
                assert out[XD + 3] == 0x00020000 and out[WD + 3] == 0x00020000
                seen.add((want["logical"], want["kind"], want["krange"], want["n0_bytes"] if tail else -1))
                unit += 1
    # every tile of the output is covered exactly once: whole, or by all its k-range pieces, or by its four column pieces
    cover = {}
    for logical, kind, krange, n0b in seen:
        cover.setdefault(logical, []).append((kind, krange, n0b))
    assert sorted(cover) == list(range(s["total"]))
    for logical, parts in cover.items():
        if len(parts) == 1:
            assert parts[0][0] == 1 and parts[0][2] == -1 and parts[0][1] == s["nk"] << 16
        elif parts[0][0] == 2:
            ks = sorted((p[1] & 0xFFFF) // 128 for p in parts)
            assert sum(p[1] >> 16 for p in parts) == s["nk"] and ks[0] == 0 and all(p[0] == 2 for p in parts)
        else:
            assert len(parts) == 4 and len({p[2] for p in parts}) == 4
