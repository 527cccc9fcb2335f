#!/usr/bin/env python3
"""Static ISA statistics of the gfx950 kernels (no GPU needed): registers, scratch, LDS and the instruction mix of each
kernel's innermost hot loop (the basic blocks between the last backward branch and its target).

  tools/isa_stats.py [--rebuild] [--filter k_rows_fast] [--loops]

Compiles astroz_amd/csrc/astroz_hip.hip with -save-temps into /tmp/az_isa (or reuses the .s there) and prints one line
per kernel.  Used to track VGPR / scratch / per-iteration instruction counts between GPU runs."""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/az_isa"
ASM = os.path.join(OUT, "astroz_hip-hip-amdgcn-amd-amdhsa-gfx950.s")


def build(defs=()):
    os.makedirs(OUT, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-save-temps",
           "-I", os.path.join(ROOT, "include"), "-o", os.path.join(OUT, "lib.so"),
           os.path.join(ROOT, "astroz_amd", "csrc", "astroz_hip.hip"), os.path.join(ROOT, "astroz_amd", "csrc", "tle_host.cpp"),
           "-x", "none", os.path.join(ROOT, "astroz_amd", "host_step.o")]  # (built by __graft_entry__.build())
    cmd += ["-D" + d for d in defs]
    subprocess.run(cmd, cwd=OUT, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.split("\n")


def classify(op):
    if op.startswith("v_pk_"):
        return "pk32"
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_fmac_f64", "v_ldexp_f64", "v_rndne_f64",
                      "v_fract_f64", "v_floor_f64", "v_trunc_f64", "v_div_", "v_frexp")):
        return "f64"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
        return "trans64"
    if op.startswith(("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp", "v_log", "v_sin", "v_cos")):
        return "trans32"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "cmp"
    if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_cndmask", "v_swap")):
        return "mov"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_store", "buffer_store", "flat_store")):
        return "vst"
    if op.startswith(("global_load", "buffer_load", "flat_load")):
        return "vld"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def parse():
    txt = open(ASM).read()
    kernels = {}
    # function bodies
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\s*\.end_amdhsa_kernel", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        kernels[name] = {"body": body}
    # metadata
    for m in re.finditer(r"\.amdhsa_kernel (_Z\w+)\n(.*?)\.end_amdhsa_kernel", txt, re.S):
        name, md = m.group(1), m.group(2)
        k = kernels.setdefault(name, {"body": ""})
        for key in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "private_segment_fixed_size", "accum_offset"):
            mm = re.search(r"\.amdhsa_%s (\d+)" % key, md)
            if mm:
                k[key] = int(mm.group(1))
    return kernels


def loops_of(body):
    """[(label, [instructions])] for every backward branch: the straight-line text between the target label and the branch."""
    lines = body.split("\n")
    labels = {}
    insts = []
    for ln in lines:
        s = ln.strip()
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        insts.append(s.split(";")[0].strip())
    out = []
    for i, ins in enumerate(insts):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", ins)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] <= i:
                out.append((tgt, insts[labels[tgt]:i + 1]))
    return out, insts


def mix(insts):
    c = collections.Counter(classify(i.split()[0]) for i in insts)
    return c


def fmt(c):
    order = ["f64", "trans64", "pk32", "trans32", "cvt", "cmp", "mov", "valu_other", "lds", "vst", "vld", "scratch", "smem", "salu", "wait"]
    valu = sum(c[k] for k in ("f64", "trans64", "pk32", "trans32", "cvt", "cmp", "mov", "valu_other"))
    return "VALU=%d [" % valu + " ".join("%s=%d" % (k, c[k]) for k in order if c[k]) + "]"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rebuild", action="store_true")
    ap.add_argument("--filter", default="k_")
    ap.add_argument("--loops", action="store_true", help="print every loop (backward branch) of the selected kernels")
    ap.add_argument("-D", action="append", default=[])
    a = ap.parse_args()
    if a.rebuild or not os.path.exists(ASM):
        build(a.D)
    ks = parse()
    names = sorted(ks)
    dem = dict(zip(names, demangle(names)))
    for n in names:
        d = dem[n].replace("void ", "").replace("(PropArgs)", "")
        if a.filter not in d:
            continue
        k = ks[n]
        lps, insts = loops_of(k["body"])
        print("%s\n    vgpr=%s sgpr=%s lds=%s scratch=%s  whole kernel: %s" % (
            d, k.get("next_free_vgpr"), k.get("next_free_sgpr"), k.get("group_segment_fixed_size"), k.get("private_segment_fixed_size"),
            fmt(mix(insts))))
        if lps:
            # the hot loop: the largest backward-branch span
            big = max(lps, key=lambda x: len(x[1]))
            print("    largest loop %s (%d insts): %s" % (big[0], len(big[1]), fmt(mix(big[1]))))
            if a.loops:
                for lab, body in lps:
                    print("      loop %s (%d insts): %s" % (lab, len(body), fmt(mix(body))))


if __name__ == "__main__":
    main()
