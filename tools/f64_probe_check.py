"""Checks tools/f64_probe's dump against CPython integers: three digit-pair squarings x <- x^2 R^-1 (mod s^2)."""
import json
import sys


def val(limbs, lb):
    return sum(int(v) << (lb * i) for i, v in enumerate(limbs))


def main(path):
    ok = True
    for line in open(path):
        d = json.loads(line)
        nl, lb = d["NL"], d["LB"]
        s = val(d["s"], lb)
        rinv = pow(1 << (lb * nl), -1, s * s)
        for k, c in enumerate(d["cases"]):
            x = (val(c["a"], lb) + val(c["b"], lb) * s) % (s * s)
            for _ in range(d["squarings"]):
                x = x * x * rinv % (s * s)
            got = (val(c["w"], lb) + val(c["v"], lb) * s) % (s * s)
            bound = max(abs(val(c["w"], lb)), abs(val(c["v"], lb))) / s
            good = got == x
            ok &= good
            print(json.dumps({"NL": nl, "limb_bits": lb, "case": k, "bit_exact": good, "max_digit_over_s": round(bound, 4)}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
