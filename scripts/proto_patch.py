"""NumPy model of the in-lane pruned transforms of stage_a2.hip (index arithmetic check)."""
import numpy as np

def W(N, m):
    return np.exp(-2j * np.pi * (np.asarray(m) % N) / N)

def dft_re_out(S, Q):
    """Re X[k1], X[k1] = sum_r S[r] W_Q^(r k1), as the kernel computes it (Q = 4 M)."""
    if Q == 2:
        return np.array([S[0].real + S[1].real, S[0].real - S[1].real])
    M = Q // 4
    V = {}
    for r1 in range(4):
        v = np.array([S[4 * r2 + r1] for r2 in range(M)])
        V[r1] = np.array([sum(v[r2] * W(M, r2 * b) for r2 in range(M)) for b in range(M)])   # dftM
        V[r1] = V[r1] * W(Q, r1 * np.arange(M))
    t0 = V[0].real + V[2].real
    t1 = V[0].real - V[2].real
    t2 = V[1].real + V[3].real
    t3 = V[1].imag - V[3].imag
    out = np.zeros(Q)
    for b in range(M):
        out[0 * M + b] = t0[b] + t2[b]
        out[2 * M + b] = t0[b] - t2[b]
        out[1 * M + b] = t1[b] + t3[b]
        out[3 * M + b] = t1[b] - t3[b]
    return out

def column(N, xin):
    """xin[n+40], n in [-40,40): Re X[x] = Re sum_n xin[n] W_N^(n x) for x in [0,N)."""
    Q = N // 64
    jmin, jmax = -((40 + Q - 1) // Q), (40 + Q - 1) // Q
    out = np.zeros(N)
    for k2 in range(64):
        S = []
        for r in range(Q):
            acc = 0
            for j in range(jmin, jmax):
                n = r + Q * j
                if -40 <= n < 40:
                    acc = acc + xin[n + 40] * (W(64, j * k2) if j else 1.0)
            S.append(acc * W(N, r * k2))
        o = dft_re_out(S, Q)
        for k1 in range(Q):
            out[64 * k1 + k2] = o[k1]
    return out

rng = np.random.default_rng(0)
for N in (128, 256, 512, 1024, 1280):
    xin = rng.normal(size=80) + 1j * rng.normal(size=80)
    full = np.zeros(N, complex)
    for n in range(-40, 40):
        full[n % N] = xin[n + 40]
    ref = np.fft.fft(full).real
    got = column(N, xin)
    print(N, np.abs(got - ref).max())
