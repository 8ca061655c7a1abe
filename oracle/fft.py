"""CPU restatement of the reference's radix-2 FFT over a prime field (TEST INFRASTRUCTURE ONLY).

Follows src/abstract/fft.ts of paulmillr/noble-curves: bit-reversal helpers :93-172,
rootsOfUnity :230-312, FFTCore :422-480, FFT :518-577.  Pinned by tests/test_oracle_golden.py
against the known-answer values of test/fft.test.ts (roots(3), brp(3), 'Basic FFT').
"""


def isPowerOfTwo(x):                       # fft.ts:56-59
    return x > 0 and (x & (x - 1)) == 0


def log2(n):                               # fft.ts:116-119
    return n.bit_length() - 1


def reverseBits(n, bits):                  # fft.ts:93-99
    r = 0
    for _ in range(bits):
        r = (r << 1) | (n & 1)
        n >>= 1
    return r


def bitReversalInplace(values):            # fft.ts:136-158
    n = len(values)
    if n < 2 or not isPowerOfTwo(n):
        raise ValueError("n must be a power of 2 and greater than 1. Got " + str(n))
    bits = log2(n)
    for i in range(n):
        j = reverseBits(i, bits)
        if i < j:
            values[i], values[j] = values[j], values[i]
    return values


def bitReversalPermutation(values):        # fft.ts:169-171
    return bitReversalInplace(list(values))


class RootsOfUnity:
    """fft.ts:230-312."""

    def __init__(self, field, generator=None):
        self.field = field
        odd = field.ORDER - 1
        p2 = 0
        while odd & 1 == 0:
            odd >>= 1
            p2 += 1
        if generator is None:              # findGenerator :175-180: smallest non-residue
            generator = 2
            while field.pow(generator, field.ORDER >> 1) == 1:
                generator += 1
        self.info = {"G": generator, "oddFactor": odd, "powerOfTwo": p2}
        self.omegas = [0] * (p2 + 1)
        self.omegas[p2] = field.pow(generator, odd)
        for i in range(p2, 0, -1):
            self.omegas[i - 1] = field.sqr(self.omegas[i])

    def _check(self, bits):
        if bits > 31 or bits > self.info["powerOfTwo"] or bits < 0:
            raise ValueError("rootsOfUnity: wrong bits %d powerOfTwo=%d" % (bits, self.info["powerOfTwo"]))
        return bits

    def roots(self, bits):                 # natural order: w^0 .. w^(N-1)
        self._check(bits)
        out, cur = [], 1
        for _ in range(1 << bits):
            out.append(cur)
            cur = self.field.mul(cur, self.omegas[bits])
        return out

    def brp(self, bits):
        return bitReversalPermutation(self.roots(bits)) if bits else self.roots(0)

    def inverse(self, bits):               # fft.ts:296-304: reversed table
        r = self.roots(bits)
        return [r[0]] + r[1:][::-1]

    def omega(self, bits):
        return self.omegas[self._check(bits)]


def FFTCore(F, N, roots, dit, invertButterflies=False, skipStages=0, brp=True):
    """fft.ts:422-480; F supplies add/sub/mul."""
    bits = log2(N)
    if not isPowerOfTwo(N):
        raise ValueError("FFT: Polynomial size should be power of two")
    if len(roots) != N:
        raise ValueError("FFT: wrong roots length: expected %d, got %d" % (N, len(roots)))
    isDit = dit != invertButterflies

    def loop(values):
        if len(values) != N:
            raise ValueError("FFT: wrong Polynomial length")
        if dit and brp and N > 1:
            bitReversalInplace(values)
        g = 1
        for i in range(bits - skipStages):
            s = i + 1 + skipStages if dit else bits - i
            m = 1 << s
            m2 = m >> 1
            stride = N >> s
            for k in range(0, N, m):
                grp = g
                g += 1
                for j in range(m2):
                    rootPos = ((N - grp) if dit else grp) if invertButterflies else j * stride
                    i0, i1 = k + j, k + j + m2
                    omega = roots[rootPos]
                    b, a = values[i1], values[i0]
                    if isDit:
                        t = F.mul(b, omega)
                        values[i0] = F.add(a, t)
                        values[i1] = F.sub(a, t)
                    elif invertButterflies:
                        values[i0] = F.add(b, a)
                        values[i1] = F.mul(F.sub(b, a), omega)
                    else:
                        values[i0] = F.add(a, b)
                        values[i1] = F.mul(F.sub(a, b), omega)
        if (not dit) and brp and N > 1:
            bitReversalInplace(values)
        return values
    return loop


class FFT:
    """fft.ts:518-577."""

    def __init__(self, roots, F):
        self.roots, self.F = roots, F

    def _loop(self, N, table, brpInput, brpOutput):
        F = self.F
        if brpInput and brpOutput:
            core = FFTCore(F, N, table, dit=False, brp=False)
            return lambda v: core(bitReversalInplace(v) if N > 1 else v)
        if brpInput:
            return FFTCore(F, N, table, dit=True, brp=False)
        if brpOutput:
            return FFTCore(F, N, table, dit=False, brp=False)
        return FFTCore(F, N, table, dit=True, brp=True)

    def direct(self, values, brpInput=False, brpOutput=False):
        N = len(values)
        if not isPowerOfTwo(N):
            raise ValueError("FFT: Polynomial size should be power of two")
        return self._loop(N, self.roots.roots(log2(N)), brpInput, brpOutput)(list(values))

    def inverse(self, values, brpInput=False, brpOutput=False):
        N = len(values)
        if not isPowerOfTwo(N):
            raise ValueError("FFT: Polynomial size should be power of two")
        res = self._loop(N, self.roots.inverse(log2(N)), brpInput, brpOutput)(list(values))
        ivm = self.F.inv(N % self.F.ORDER)
        return [self.F.mul(x, ivm) for x in res]
