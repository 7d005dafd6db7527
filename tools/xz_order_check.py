"""The division-free order of the "x/z" strings of extensionAligner's std::set<std::string> achieved_complete_sequence_alignments (extensionAligner.cpp:493, 1431) used by
kernel_dp.hip: xz_less -- pad the shorter number with zeros, compare as numbers, a proper prefix first -- against Python's string order (= std::string's operator<: bytewise)."""
import random


def part(a, b):
    la, lb = len(str(a)), len(str(b)); L = max(la, lb)
    A = a * 10 ** (L - la); B = b * 10 ** (L - lb)
    if A != B:
        return -1 if A < B else 1
    if la != lb:
        return -1 if la < lb else 1
    return 0


def fast(x1, z1, x2, z2):
    c = part(x1, x2)
    return c < 0 if c else part(z1, z2) < 0


if __name__ == "__main__":
    random.seed(1)
    vals = [0, 1, 9, 10, 11, 12, 19, 99, 100, 101, 119, 120, 121, 123, 125, 129, 130, 999, 1000, 1200, 1234, 12345, 99999, 100000, 1199999, 1200000, 16777215]
    zs = (0, 1, 5, 9, 10, 51, 99, 100, 477, 1000, 99999)
    n = 0
    for x1 in vals:
        for x2 in vals:
            for z1 in zs:
                for z2 in zs:
                    assert (f"{x1}/{z1}" < f"{x2}/{z2}") == fast(x1, z1, x2, z2), (x1, z1, x2, z2); n += 1
    for _ in range(1000000):
        k = random.choice([1, 2, 3, 4, 5, 6, 7, 8]); x1 = random.randrange(10 ** k if k < 8 else 16777216)
        x2 = max(0, min(16777215, random.choice([x1, x1 + random.randrange(-300, 300), random.randrange(16777216), x1 * 10, x1 // 10])))
        z1 = random.randrange(random.choice([10, 100, 1000, 100000])); z2 = random.choice([z1, random.randrange(100000), z1 * 10 % 100000, z1 // 10])
        assert (f"{x1}/{z1}" < f"{x2}/{z2}") == fast(x1, z1, x2, z2), (x1, z1, x2, z2); n += 1
    print("xz order: %d pairs agree with the string order" % n)
