"""Design check (numpy, float64) of the data flow of csrc/embed_fft.inc: a 400-point real STFT frame as a packed 200-point complex
FFT spread over 8 lanes -- 25-point FFTs in registers, a radix-8 step across the lanes through DPP-shaped exchanges (mirror, xor 2,
xor 1), the real-input split against the mirrored lane.  Prints the lane tables the kernel needs and checks everything against
numpy.fft.rfft.  Not part of the product; the tables themselves are rebuilt in C++ by radad_embed_create."""
import numpy as np

N, NC = 400, 200


def r_of_lane(p):
    return p if p < 4 else 11 - p


def fft25(u):
    """5 x 5 Cooley-Tukey exactly as the kernel does it: m = 5 m1 + m2, k1 = j1 + 5 j2"""
    W5 = np.exp(-2j * np.pi / 5)
    W25 = np.exp(-2j * np.pi / 25)
    V = np.zeros((5, 5), complex)          # [m2][j1]
    for m2 in range(5):
        x = u[m2::5]
        for j1 in range(5):
            V[m2, j1] = sum(x[m1] * W5 ** (m1 * j1) for m1 in range(5)) * W25 ** (m2 * j1)
    Y = np.zeros(25, complex)
    for j1 in range(5):
        for j2 in range(5):
            Y[j1 + 5 * j2] = sum(V[m2, j1] * W5 ** (m2 * j2) for m2 in range(5))
    return Y


def network(T, tables=None):
    """T[p][k1]: lane p holds the twiddled 25-point result of residue r_of_lane(p).  Three exchange stages (partner 7 - p, p ^ 2,
    p ^ 1); the lane whose bit (2, 1, 0) of p is clear keeps the sum, the other the twiddled difference:
        out = (partner + beta * own) * tw          beta = +-1, tw complex, both per (stage, lane)
    so that lane p ends with block k2 = bitrev3(p) and lane 7 - p with block 7 - k2."""
    W8 = np.exp(-2j * np.pi / 8)
    x = T.copy()
    idx = [r_of_lane(p) for p in range(8)]
    for stage, (partner, half, bit) in enumerate(((lambda p: 7 - p, 4, 2), (lambda p: p ^ 2, 2, 1), (lambda p: p ^ 1, 1, 0))):
        y = np.zeros_like(x)
        nidx = list(idx)
        for p in range(8):
            q = partner(p)
            small = idx[p] < half
            assert (idx[q] < half) != small and idx[q] % half == idx[p] % half
            hi = (p >> bit) & 1
            if not hi:
                beta, tw = 1.0, 1.0 + 0j
            else:
                # (x_small - x_big) W^(small index): own is small -> own - partner = -(partner - own)
                e = (idx[p] % half) * (4 // half)
                tw = W8 ** e
                beta = -1.0
                if small:
                    tw = -tw
            y[p] = (x[q] + beta * x[p]) * tw
            nidx[p] = idx[p] % half
            if tables is not None:
                tables.append((stage, p, beta, tw))
        x, idx = y, nidx
    return x


def main():
    rng = np.random.default_rng(0)
    yv = rng.standard_normal(N)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)
    g = w * yv
    z = g[0::2] + 1j * g[1::2]
    Zref = np.fft.fft(z)
    Xref = np.fft.rfft(g)
    T = np.zeros((8, 25), complex)
    for p in range(8):
        r = r_of_lane(p)
        u = z[r::8]
        Y = fft25(u)
        assert np.allclose(Y, np.fft.fft(u))
        T[p] = Y * np.exp(-2j * np.pi * r * np.arange(25) / 200)
    out = network(T)
    k2 = []
    for p in range(8):
        found = [c for c in range(8) if np.allclose(out[p], Zref[25 * c:25 * c + 25])]
        assert len(found) == 1, (p, found)
        k2.append(found[0])
    print("k2 of lane:", k2)
    print("mirror property k2[7-p] == 7-k2[p]:", all(k2[7 - p] == 7 - k2[p] for p in range(8)))
    lane_of = {k2[p]: p for p in range(8)}
    print("k1=0 partner lane (holds (8-k2)%8):", [lane_of[(8 - k2[p]) % 8] for p in range(8)])
    # real split
    X = np.zeros(201, complex)
    for p in range(8):
        c = k2[p]
        for k1 in range(25):
            k = 25 * c + k1
            A = out[p][k1]
            if k1 == 0:
                B = np.conj(out[lane_of[(8 - c) % 8]][0])
            else:
                B = np.conj(out[7 - p][25 - k1])
            E, O = A + B, A - B
            X[k] = 0.5 * (E - 1j * np.exp(-2j * np.pi * k / N) * O)
    X[200] = (out[lane_of[0]][0].real - out[lane_of[0]][0].imag)
    print("max |X - rfft|:", np.abs(X - Xref).max())


if __name__ == "__main__":
    main()
