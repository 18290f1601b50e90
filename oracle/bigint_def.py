"""ORACLE — TEST INFRASTRUCTURE ONLY.

Python big-integer DEFINITIONS used to pin the C restatement (oracle/lw_oracle.c) independently of its
algorithmic structure: the NTT as the defining sum, field ops as `% p`, the group law in affine
coordinates, MSM as the affine sum of k_i * P_i.  Pure-Python loops: small cases only.

Constants are transcribed from the reference (file:line beside each).
"""

# ---- moduli ---------------------------------------------------------------
P_STARK252 = 0x800000000000011000000000000000000000000000000000000000000000001  # stark_252_prime_field.rs:13-14
P_FR381 = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001   # bls12_381/default_types.rs:15-17
P_BABYBEAR = 2013265921                                                        # babybear.rs:16, babybear_u32.rs:6
P_FP381 = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab  # bls12_381/field_extension.rs:13
P_FP254 = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47   # bn_254/field_extension.rs:15-16
P_FR254 = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001   # bn_254/default_types.rs:14-16

# (TWO_ADICITY, TWO_ADIC_PRIMITVE_ROOT_OF_UNITY)
FFT_PARAMS = {
    P_STARK252: (192, 0x5282db87529cfa3f0464519c8b0fa5ad187148e11a61616070024f42f8ef94),   # stark_252_prime_field.rs:19-24
    P_FR381: (32, 0x2ab00961a08a499d84dd396c349d9b3cc5e433d6fa78eb2b54cc39d9bb30bbb7),     # bls12_381/default_types.rs:25-30
    P_BABYBEAR: (24, 21),                                                                # babybear.rs:28-31
}


def primitive_root_of_unity(p, order):
    """traits.rs:82-94 as plain modular arithmetic."""
    ta, g = FFT_PARAMS[p]
    if order == 0:
        return 1
    if order > ta:
        raise ValueError("RootOfUnityError")
    return pow(g, 1 << (ta - order), p)


def ntt_by_definition(coeffs, p, w):
    n = len(coeffs)
    return [sum(c * pow(w, (i * j) % n, p) for j, c in enumerate(coeffs)) % p for i in range(n)]


def evaluate_fft_def(coeffs, p, blowup=1, domain_size=None, offset=None):
    """Polynomial::evaluate_fft semantics (fft/polynomial.rs:25-38,74-82) by Horner evaluation."""
    c = list(coeffs)
    while c and c[-1] % p == 0:
        c.pop()
    ds = domain_size or 0
    m = max(len(c), ds)
    length = 1
    while length < m:
        length <<= 1
    length *= blowup
    if not c:
        return [0] * length
    order = length.bit_length() - 1
    assert 1 << order == length
    w = primitive_root_of_unity(p, order)
    h = 1 if offset is None else offset
    out = []
    for i in range(length):
        x = h * pow(w, i, p) % p
        acc = 0
        for cj in reversed(c):
            acc = (acc * x + cj) % p
        out.append(acc)
    return out


def interpolate_fft_def(evals, p, offset=None):
    """Inverse DFT by definition, then divide out the coset powers."""
    n = len(evals)
    order = n.bit_length() - 1
    w = primitive_root_of_unity(p, order)
    winv = pow(w, -1, p)
    ninv = pow(n, -1, p)
    c = [v * ninv % p for v in ntt_by_definition(evals, p, winv)]
    if offset is not None:
        hinv = pow(offset, -1, p)
        c = [v * pow(hinv, i, p) % p for i, v in enumerate(c)]
    return c


def bit_reverse(i, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


# ---- Fp2 = Fp[u]/(u^2+1) (both towers) -------------------------------------
class Fp2:
    __slots__ = ("p", "a", "b")

    def __init__(self, p, a, b=0):
        self.p, self.a, self.b = p, a % p, b % p

    def _c(self, o):
        return o if isinstance(o, Fp2) else Fp2(self.p, o, 0)

    def __add__(self, o):
        o = self._c(o); return Fp2(self.p, self.a + o.a, self.b + o.b)
    __radd__ = __add__

    def __sub__(self, o):
        o = self._c(o); return Fp2(self.p, self.a - o.a, self.b - o.b)

    def __rsub__(self, o):
        return self._c(o) - self

    def __neg__(self):
        return Fp2(self.p, -self.a, -self.b)

    def __mul__(self, o):
        o = self._c(o)
        return Fp2(self.p, self.a * o.a - self.b * o.b, self.a * o.b + self.b * o.a)
    __rmul__ = __mul__

    def inv(self):
        n = pow(self.a * self.a + self.b * self.b, -1, self.p)
        return Fp2(self.p, self.a * n, -self.b * n)

    def __eq__(self, o):
        o = self._c(o); return self.a == o.a and self.b == o.b

    def __hash__(self):
        return hash((self.a, self.b))

    def is_zero(self):
        return self.a == 0 and self.b == 0

    def tup(self):
        return (self.a, self.b)


class Fp:
    __slots__ = ("p", "v")

    def __init__(self, p, v):
        self.p, self.v = p, v % p

    def _c(self, o):
        return o if isinstance(o, Fp) else Fp(self.p, o)

    def __add__(self, o):
        return Fp(self.p, self.v + self._c(o).v)
    __radd__ = __add__

    def __sub__(self, o):
        return Fp(self.p, self.v - self._c(o).v)

    def __rsub__(self, o):
        return self._c(o) - self

    def __neg__(self):
        return Fp(self.p, -self.v)

    def __mul__(self, o):
        return Fp(self.p, self.v * self._c(o).v)
    __rmul__ = __mul__

    def inv(self):
        return Fp(self.p, pow(self.v, -1, self.p))

    def __eq__(self, o):
        return self.v == self._c(o).v

    def __hash__(self):
        return hash(self.v)

    def is_zero(self):
        return self.v == 0

    def tup(self):
        return self.v


# ---- curves: y^2 = x^3 + b, affine law, None = point at infinity -----------
class Curve:
    def __init__(self, name, p, b, gen, fp2=False):
        self.name, self.p, self.fp2 = name, p, fp2
        self.F = (lambda *a: Fp2(p, *a)) if fp2 else (lambda a: Fp(p, a))
        self.b = self.F(*b) if fp2 else self.F(b)
        self.gen = self.pt(*gen)

    def pt(self, x, y):
        if self.fp2:
            return (self.F(*x), self.F(*y))
        return (self.F(x), self.F(y))

    def on_curve(self, P):
        if P is None:
            return True
        x, y = P
        return y * y == x * x * x + self.b

    def add(self, P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if y1 == y2 and not y1.is_zero():
                lam = (x1 * x1 * 3) * (y1 * 2).inv()
            else:
                return None
        else:
            lam = (y2 - y1) * (x2 - x1).inv()
        x3 = lam * lam - x1 - x2
        y3 = lam * (x1 - x3) - y1
        return (x3, y3)

    def neg(self, P):
        return None if P is None else (P[0], -P[1])

    def mul(self, k, P):
        R = None
        while k:
            if k & 1:
                R = self.add(R, P)
            P = self.add(P, P)
            k >>= 1
        return R

    def msm(self, ks, Ps):
        R = None
        for k, P in zip(ks, Ps):
            R = self.add(R, self.mul(k, P))
        return R

    def tup(self, P):
        return None if P is None else (P[0].tup(), P[1].tup())


# generators: bls12_381/curve.rs:29-35; bn_254/curve.rs:23-29; bn_254/twist.rs:10-21; bls12_381/twist.rs:11-14
BLS12_381_G1 = Curve(
    "bls12_381_g1", P_FP381, 4,
    (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
     0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1))
BN254_G1 = Curve("bn254_g1", P_FP254, 3, (1, 2))
BN254_G2 = Curve(
    "bn254_g2", P_FP254,
    (0x2b149d40ceb8aaae81be18991be06ac3b5b4c5e559dbefa33267e6dc24a138e5,
     0x009713b03af0fed4cd2cafadeed8fdf4a74fa084e52d1852e4a2bd0685c315d2),   # bn_254/twist.rs:46-61
    ((0x1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed,
      0x198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2),
     (0x12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa,
      0x090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b)), fp2=True)
BLS12_381_G2 = Curve(
    "bls12_381_g2", P_FP381, (4, 4),                                           # bls12_381/twist.rs:40-48
    ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
      0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
     (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
      0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)), fp2=True)

CURVES = [BLS12_381_G1, BN254_G1, BN254_G2, BLS12_381_G2]   # index = ORC_C_* / LW_CURVE_* id


def adds_ref(n, num_limbs=4):
    """Point additions the reference's msm() performs for n points (pippenger.rs:34-40,51-57,66-98):
    n*W scatter adds + 2*W*(2^c - 1) running-sum adds."""
    lg = n.bit_length() - 1 if n else 0
    c = min(max((lg * 4) // 5, 2), 32)
    W = (64 * num_limbs - 1) // c + 1
    return n * W + 2 * W * ((1 << c) - 1)
