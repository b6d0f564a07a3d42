"""Counts the VALU mnemonics of every microbenchmark kernel in valu_rates.s, so that a row's
name can be checked against what the compiler emitted for it."""
import collections
import re
import sys

NAMES = ["v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_mul_f32(sgpr)",
         "v_min3_f32", "v_max_f32", "v_cmp_lt_f32", "v_cndmask_b32", "v_readlane_b32", "v_rcp_f32", "v_sqrt_f32",
         "mix pk_mul:pk_add 1:1", "mix cmp:cndmask 1:1", "mix fma:readlane 3:1"]


def main(path):
    text = open(path).read()
    parts = re.split(r"\n(_Z1kILi(\d+)EEvPfiff):[^\n]*\n", text)
    print("instructions in the kernel's code (loop body = 128 of the named kind; the rest is set-up and the final sum):")
    for i in range(1, len(parts), 3):
        kind = int(parts[i + 1])
        body = parts[i + 2].split("s_endpgm")[0]
        c = collections.Counter(re.findall(r"^\s+([vs]_[a-z0-9_]+)", body, re.M))
        valu = {k: v for k, v in c.items() if k.startswith("v_")}
        top = ", ".join(f"{k} {v}" for k, v in sorted(valu.items(), key=lambda kv: -kv[1])[:5])
        print(f"  {NAMES[kind]:<22} {top}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "valu_rates.s")
