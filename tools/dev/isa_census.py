"""dev: instruction census of a kernel's loops from the gfx950 ISA (runs without a GPU).

usage: python tools/dev/isa_census.py fealess_amd/csrc/fl_icp.hip k_icp_pipelineILi0ELi256ELi4 [extra hipcc flags]

Compiles the file with the library's flags to assembly, takes the function whose mangled name contains the given text, finds
its loops (back edges between basic blocks) and prints for each: blocks, instructions, vector / scalar / LDS / vector-memory
counts and the quarter-rate or multi-pass ones by name (32-bit integer multiplies, 64-bit multiply-adds, fp64, divisions,
transcendentals, packed float32).  Static counts: multiply by the trip counts yourself.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize".split()
SLOW = ("v_mul_lo_u32", "v_mul_lo_i32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32", "v_rcp_", "v_rsq_", "v_sqrt_",
        "v_div_", "v_exp_", "v_log_", "v_sin_", "v_cos_", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "_f64")


def main():
    src, want = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *sys.argv[3:], "-S", "--cuda-device-only", "-o", out, src], check=True,
                       stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z]\w*:", l) and want in l.split(":")[0]]
    if not starts:
        sys.exit("no function matching %r" % want)
    a = starts[0]
    b = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks = [{"name": "entry", "ops": [], "br": []}]
    for l in lines[a + 1:b]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append({"name": m.group(1), "ops": [], "br": []})
            continue
        t = l.strip()
        if not t or t[0] in ";.":
            continue
        op = t.split()[0]
        blocks[-1]["ops"].append(op)
        if op.startswith("s_cbranch") or op == "s_branch":
            blocks[-1]["br"].append(t.split()[-1])
    idx = {blk["name"]: k for k, blk in enumerate(blocks)}
    print("%s: %d instructions in %d blocks" % (lines[a].split(":")[0], sum(len(x["ops"]) for x in blocks), len(blocks)))
    seen = set()
    for k, blk in enumerate(blocks):
        for t in blk["br"]:
            if t not in idx or idx[t] > k or (t, blk["name"]) in seen:
                continue
            seen.add((t, blk["name"]))
            ops = [o for x in blocks[idx[t]:k + 1] for o in x["ops"]]
            c = collections.Counter(ops)
            cls = lambda p: sum(v for o, v in c.items() if o.startswith(p))
            slow = {o: v for o, v in c.items() if any(s in o for s in SLOW)}
            print("loop %-12s .. %-12s blocks %3d  n %5d  valu %5d  salu %5d  lds %4d  vmem %4d  %s" %
                  (t, blk["name"], k + 1 - idx[t], len(ops), cls("v_"), cls("s_"), cls("ds_"),
                   cls("global_") + cls("buffer_") + cls("scratch_") + cls("flat_"), slow if slow else ""))


if __name__ == "__main__":
    main()
