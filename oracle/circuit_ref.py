"""Noise-free integer evaluation of a compiled circuit blob, in numpy, on the phase domain.

TEST INFRASTRUCTURE ONLY (see oracle/tfhe_ref.h).  This is the "noise-free integer circuit" that
SURVEY.md section 8(c) P1 defines parity against: what the reference calls the quantised / clear
forward of `QuantizedModule.forward` (homomorphic_eval.py:70 with fhe="disable"/"simulate" [K]).
It parses the blob on its own (format: DESIGN.md section 4) and shares no code with the product.
A ciphertext is replaced by its plaintext phase (one u64 word); every operator is the same
wrap-around arithmetic the encrypted operator performs on phases.
"""
import struct

import numpy as np

OP_CONV, OP_ADD, OP_SUMPOOL, OP_LUT = 1, 2, 3, 4


def parse_blob(blob):
    magic, ver, nT, nO, tin, tout, maxbits, _ = struct.unpack_from("<IIiiiiii", blob, 0)
    assert magic == 0x46544344 and ver == 1
    off = 32
    tensors = [struct.unpack_from("<iiii", blob, off + 16 * i)[:3] for i in range(nT)]
    off += 16 * nT
    ops = []
    for i in range(nO):
        rec = struct.unpack_from("<iiii12i2qqq", blob, off + 96 * i)
        typ, s0, s1, dst = rec[:4]
        ip, lp, poff, plen = rec[4:16], rec[16:18], rec[18], rec[19]
        ops.append(dict(type=typ, src0=s0, src1=s1, dst=dst, ip=ip, lp=lp, payload=blob[poff:poff + plen] if plen else b""))
    return dict(tensors=tensors, ops=ops, input=tin, output=tout, max_bit_width=maxbits)


def _conv_u64(x, w, stride, pad):
    """x [B,Cin,H,W] uint64 phases, w [Cout,Cin,KH,KW] int8 -> wrap-around conv"""
    B, Cin, H, W = x.shape
    Cout, _, KH, KW = w.shape
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    xp = np.zeros((B, Cin, H + 2 * pad, W + 2 * pad), np.uint64)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    wu = w.astype(np.int64).astype(np.uint64)
    out = np.zeros((B, Cout, Ho, Wo), np.uint64)
    for ky in range(KH):
        for kx in range(KW):
            patch = xp[:, :, ky:ky + (Ho - 1) * stride + 1:stride, kx:kx + (Wo - 1) * stride + 1:stride]   # [B,Cin,Ho,Wo]
            out += np.einsum("oc,bcyx->boyx", wu[:, :, ky, kx], patch)
    return out


def run_clear(blob, phases_in):
    """phases_in: uint64 [B, n_in] -> (uint64 [B, n_out], overflow flag)"""
    c = parse_blob(blob)
    T = c["tensors"]
    B = phases_in.shape[0]
    vals = {c["input"]: np.ascontiguousarray(phases_in, np.uint64).reshape(B, *T[c["input"]])}
    overflow = False
    for o in c["ops"]:
        x = vals[o["src0"]]
        ip = o["ip"]
        if o["type"] == OP_CONV:
            Cout, KH, KW, stride, pad = ip[:5]
            w = np.frombuffer(o["payload"], np.int8).reshape(Cout, x.shape[1], KH, KW)
            y = _conv_u64(x, w, stride, pad)
        elif o["type"] == OP_ADD:
            y = x + vals[o["src1"]]
        elif o["type"] == OP_SUMPOOL:
            K = ip[0]
            Ho, Wo = x.shape[2] // K, x.shape[3] // K
            y = x[:, :, :Ho * K, :Wo * K].reshape(B, x.shape[1], Ho, K, Wo, K).sum(axis=(3, 5), dtype=np.uint64)
        elif o["type"] == OP_LUT:
            p, r, w, shift, _, _, ntab = ip[:7]
            tables = np.frombuffer(o["payload"], np.int64).reshape(ntab, 1 << w).view(np.uint64)
            v = (x << np.uint64(shift)) + np.uint64(o["lp"][0] % (1 << 64))
            if r > 0:
                v = v + (np.uint64(1) << np.uint64(63 - p + r - 1))
            overflow |= bool((v >> np.uint64(63)).any())
            idx = ((v >> np.uint64(63 - w)) & np.uint64((1 << w) - 1)).astype(np.int64)
            ch = np.arange(x.shape[1]).reshape(1, -1, 1, 1) if ntab > 1 else np.zeros((1, 1, 1, 1), np.int64)
            y = tables[np.broadcast_to(ch, idx.shape), idx]
        else:
            raise ValueError("unknown op")
        vals[o["dst"]] = y
    return vals[c["output"]].reshape(B, -1), overflow
