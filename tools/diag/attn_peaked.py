"""Diagnostic: per-row error of fg_attn_fwd_bf16 on peaked / spiked inputs (which rows, which kernels).
    python tools/diag/attn_peaked.py [lib.so]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fairygen_amd import hip

lib = hip.load()
if len(sys.argv) > 1:
    hip._lib = None
    alt = ctypes.CDLL(os.path.abspath(sys.argv[1]))
    for name, sig in hip._SIGNATURES.items():
        if hasattr(alt, name):
            getattr(alt, name).argtypes = sig
            getattr(alt, name).restype = ctypes.c_int
    alt.fg_last_error.restype = ctypes.c_char_p
    alt.fg_attn_workspace_bytes.restype = ctypes.c_int64
    lib = alt


def seeded(shape, seed, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float32) * scale).to(torch.bfloat16)


def run(q, k, v, heads):
    b, nq, c = q.shape
    nkv = k.shape[1]
    out = torch.zeros_like(q)
    need = lib.fg_attn_workspace_bytes(b, nq, nkv, heads)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    rc = lib.fg_attn_fwd_bf16(q.data_ptr(), c, k.data_ptr(), c, v.data_ptr(), c, out.data_ptr(), b, nq, nkv, heads, 128, 128 ** -0.5,
                              ws.data_ptr() if need else None, need, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.fg_last_error()
    torch.cuda.synchronize()
    return out


def ref(q, k, v, heads, dt):
    b, nq, c = q.shape
    qh = q[0].view(nq, heads, 128).to(dt).transpose(0, 1)
    kh = k[0].view(-1, heads, 128).to(dt).transpose(0, 1)
    vh = v[0].view(-1, heads, 128).to(dt).transpose(0, 1)
    s = (qh @ kh.transpose(1, 2)).float() * 128 ** -0.5
    p = torch.softmax(s, dim=-1).to(dt)
    return (p @ vh).transpose(0, 1).reshape(nq, c).float()


for nq, nkv, heads, sq, spikes in [(300, 1500, 2, 1.0, False), (300, 1500, 2, 8.0, False), (300, 1500, 2, 8.0, True), (300, 1500, 2, 3.0, False),
                                   (300, 1472, 2, 8.0, False), (2000, 1500, 24, 8.0, False), (256, 1088, 1, 8.0, False), (256, 1088, 1, 4.0, False)]:
    c = heads * 128
    q, k, v = seeded((1, nq, c), 130, sq).cuda(), seeded((1, nkv, c), 131).cuda(), seeded((1, nkv, c), 132).cuda()
    if spikes:
        for i, (pos, row) in enumerate([(70, 5), (nkv // 3, 17), (nkv // 2, 150), ((nkv // 64) * 64 - 3, 255), (nkv - 2, 299), (nkv - 1, 5)]):
            k[0, pos] = (q[0, row].float() * (1.5 + i)).to(torch.bfloat16)
    R, S = ctypes.c_int(), ctypes.c_int()
    hip.load().fg_attn_split_choice(1, nq, nkv, heads, hip.load().fg_attn_workspace_bytes(1, nq, nkv, heads), ctypes.byref(R), ctypes.byref(S))
    got = run(q, k, v, heads)[0].float()
    r32, r16 = ref(q, k, v, heads, torch.float32), ref(q, k, v, heads, torch.bfloat16)
    err = (got - r32).abs().view(nq, heads, 128).amax(-1)          # (row, head)
    e16 = (r16 - r32).abs().max().item()
    bad = (err > 2 * e16 + 2e-3).nonzero()
    print(f"nq={nq} nkv={nkv} H={heads} scale_q={sq} spikes={spikes} split(R={R.value},S={S.value}): max err {err.max().item():.4f} (ref16 {e16:.4f}); "
          f"{len(bad)} bad (row, head) of {nq * heads}; first: {bad[:12].tolist()}", flush=True)
    if len(bad):
        rows = bad[:, 0]
        print("   bad rows mod 64 histogram:", torch.bincount(rows % 64, minlength=64).tolist())
        print("   bad rows // 32 histogram:", torch.bincount(rows // 32).tolist())
        r, h = bad[0].tolist()
        s = (q[0, r, h * 128:(h + 1) * 128].float() @ k[0, :, h * 128:(h + 1) * 128].float().T) * 128 ** -0.5 * 1.4427
        top = s.topk(4)
        tile_max = s[: (nkv // 64) * 64].view(-1, 64).amax(-1)
        print(f"   row {r} head {h}: top log2-scores {top.values.tolist()} at keys {top.indices.tolist()}; running-max jumps (tile, new max):",
              [(i, round(tile_max[i].item(), 1)) for i in range(len(tile_max)) if tile_max[i] > tile_max[:i].max(initial=-1e9) if i == 0 or True][:12]
              if False else [(i, round(float(tile_max[i]), 1)) for i in range(len(tile_max)) if i == 0 or tile_max[i] > tile_max[:i].max()])
