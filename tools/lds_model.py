"""LDS bank-conflict model of the FFT exchange patterns (fft_core.h) under the gfx950 banking rules of
/opt/skills/guides/MI355X_MICROARCH.md (section LDS): a wave's ds_read_b128 is served in 4 groups of 16 lanes over 64 banks of 4 bytes,
its ds_write_b128 in 8 groups of 8 contiguous lanes over 32 banks; inside a group every extra distinct address on a bank costs a cycle.
Prints, per exchange pattern and candidate index skew, the cycles per wave-instruction (ideal: 4 for a read, 8 for a write).
usage: lds_model.py [logM] [P]"""
import sys

READ_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
READ_GROUPS += [[l + 32 for l in g] for g in READ_GROUPS]
WRITE_GROUPS = [list(range(8 * g, 8 * g + 8)) for g in range(8)]


def cycles(byte_addrs, groups, nbanks):
    tot = 0
    for g in groups:
        per_bank = {}
        for l in g:
            a = byte_addrs[l]
            for b in range(4):
                per_bank.setdefault(((a // 4) + b) % nbanks, set()).add(a // 4 + b)
        tot += max(len(v) for v in per_bank.values())
    return tot


def passes(logM, P):
    lp = {4: 2, 8: 3, 16: 4}[P]
    S = (logM + lp - 1) // lp
    R = [P] * S
    if logM % lp:
        R[-1] = 1 << (logM % lp)
    W = [1] * S
    for i in range(S - 2, -1, -1):
        W[i] = W[i + 1] * R[i + 1]
    return R, W


def addr(logM, P, i, t, j):
    R, W = passes(logM, P)
    if R[i] == P:
        return (t // W[i]) * (W[i] * R[i]) + t % W[i] + j * W[i]
    return P * t + j


SKEWS = {
    "idx + idx/8 (shipped)": lambda i: i + (i >> 3),
    "none": lambda i: i,
    "idx + idx/16": lambda i: i + (i >> 4),
    "idx + idx/32": lambda i: i + (i >> 5),
    "idx + idx/64": lambda i: i + (i >> 6),
    "idx ^ ((idx>>3)&7)": lambda i: i ^ ((i >> 3) & 7),
    "idx ^ ((idx>>4)&7)": lambda i: i ^ ((i >> 4) & 7),
    "idx ^ ((idx>>3)&3) + idx/64": lambda i: (i ^ ((i >> 3) & 3)) + (i >> 6),
    "(idx ^ ((idx>>3)&7)) + idx/64": lambda i: (i ^ ((i >> 3) & 7)) + (i >> 6),
    "(idx ^ ((idx>>6)&7)) + idx/8": lambda i: (i ^ ((i >> 6) & 7)) + (i >> 3),
    "idx + idx/8 + idx/64": lambda i: i + (i >> 3) + (i >> 6),
    "idx + 2*(idx/16)": lambda i: i + 2 * (i >> 4),
    "idx + idx/4": lambda i: i + (i >> 2),
}


def main():
    logM = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    R, W = passes(logM, P)
    T = (1 << logM) // P
    print(f"M = 2^{logM}, P = {P}, T = {T}: radices {R}, weights {W}")
    for name, sk in SKEWS.items():
        row, total = [], 0
        for i in range(len(R)):
            rd = wr = 0
            nw = max(1, T // 64)
            for w in range(min(nw, 8)):            # a few waves: the pattern repeats
                for j in range(P):
                    a = [16 * sk(addr(logM, P, i, 64 * w + l if T >= 64 else l % T, j)) for l in range(64)]
                    rd += cycles(a, READ_GROUPS, 64)
                    wr += cycles(a, WRITE_GROUPS, 32)
            n = min(nw, 8) * P
            row.append(f"A{i}: read {rd / n:5.2f} write {wr / n:5.2f}")
            # every pattern is read once and written once per forward + inverse pair, except A0 (write fwd, read inv) and A_last (read fwd, write inv): same counts
            total += rd / n + wr / n
        print(f"{name:34s} " + " | ".join(row) + f" | sum {total:6.2f} (ideal {12 * len(R)})")


if __name__ == "__main__":
    main()
