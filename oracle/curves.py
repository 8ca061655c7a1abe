"""Curve instances on the hot path: constants restated from the reference's
`src/secp256k1.ts:48-64`, `src/ed25519.ts:49-65`, `src/bls12-381.ts:134-158,321-345`.
Test infrastructure - see oracle/__init__.py.
"""
from .edwards import ED25519_P, ed25519_uvRatio, edwards
from .field import Field, Field2
from .weierstrass import weierstrass

# ---------------------------------------------------------------- secp256k1
SECP256K1_P = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F
SECP256K1_N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
secp256k1_CURVE = dict(
    p=SECP256K1_P, n=SECP256K1_N, h=1, a=0, b=7,
    Gx=0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
    Gy=0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8,
)
secp256k1_ENDO = dict(                                  # secp256k1.ts:58-64
    beta=0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE,
    basises=(
        (0x3086D221A7D46BCDE86C90E49284EB15, -0xE4437ED6010E88286F547FA90ABFE4C3),
        (0x114CA50F7A8E2F3F657C1108D9D44CFD8, 0x3086D221A7D46BCDE86C90E49284EB15),
    ),
)
Fp_k1 = Field(SECP256K1_P)
Fn_k1 = Field(SECP256K1_N)
Secp256k1 = weierstrass(secp256k1_CURVE, Fp_k1, Fn_k1, endo=secp256k1_ENDO, name="secp256k1")

# ---------------------------------------------------------------- ed25519
ED25519_L = 0x1000000000000000000000000000000014DEF9DEA2F79CD65812631A5CF5D3ED
ed25519_CURVE = dict(
    p=ED25519_P, n=ED25519_L, h=8,
    a=ED25519_P - 1,                                     # ed25519.ts:60 (a = -1)
    d=0x52036CEE2B6FFE738CC740797779E89800700A4D4141D8AB75EB4DCA135978A3,
    Gx=0x216936D3CD6E53FEC0A4E231FDD6DC5C692CC7609525A7B2C9562D608F25D51A,
    Gy=0x6666666666666666666666666666666666666666666666666666666666666658,
)
Fp_25519 = Field(ED25519_P, is_le=True)                 # curve.ts:1036 (isLE for Edwards)
Fn_25519 = Field(ED25519_L, is_le=True)
Ed25519 = edwards(ed25519_CURVE, Fp_25519, Fn_25519, ed25519_uvRatio, name="ed25519")

# ---------------------------------------------------------------- bls12-381
BLS_P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
BLS_R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
Fp_bls = Field(BLS_P)
Fr_bls = Field(BLS_R)
Fp2_bls = Field2(Fp_bls)
bls_G1_CURVE = dict(
    p=BLS_P, n=BLS_R, h=0x396C8C005555E1568C00AAAB0000AAAB, a=0, b=4,
    Gx=0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    Gy=0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
)
bls_G2_CURVE = dict(
    p=Fp2_bls.ORDER, n=BLS_R,
    h=0x5D543A95414E7F1091D50792876A202CD91DE4547085ABAA68A205B2E5A7DDFA628F1CB4D9E82EF21537E293A6691AE1616EC6E786F0C70CF1C38E31C7238E5,
    a=(0, 0), b=(4, 4),
    Gx=(0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
        0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
    Gy=(0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
        0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
)
BlsG1 = weierstrass(bls_G1_CURVE, Fp_bls, Fr_bls, name="bls12_381_G1")
BlsG2 = weierstrass(bls_G2_CURVE, Fp2_bls, Fr_bls, name="bls12_381_G2")

CURVES = {"secp256k1": Secp256k1, "ed25519": Ed25519, "bls12_381_g1": BlsG1, "bls12_381_g2": BlsG2}


class makeRng:
    """xorshift64 generator of the reference's soak tests (test/point.test.ts:536-559),
    same rnd64 / rndBig / rndBelow so synthetic inputs are reproducible across both."""
    M = 0xFFFFFFFFFFFFFFFF

    def __init__(self, seed):
        self.seed = seed

    def rnd64(self):
        M = self.M
        s = self.seed
        s = (s ^ (s << 13)) & M
        s ^= s >> 7
        s = (s ^ (s << 17)) & M
        self.seed = s
        return s

    def rndBig(self, bits):
        r = 0
        for _ in range(0, bits, 64):
            r = (r << 64) | self.rnd64()
        return r & ((1 << bits) - 1)

    def rndBelow(self, n):
        bits = n.bit_length()
        while True:
            r = self.rndBig(bits)
            if r < n:
                return r
