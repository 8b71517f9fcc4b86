#!/usr/bin/env python3
"""Compile csrc/kernels.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per
kernel: VGPRs, AGPRs, SGPRs, scratch bytes, occupancy (waves per SIMD), LDS.  Optional: --isa writes the
disassembly to /tmp/kernels.s and prints VALU / SALU / VMEM / LDS instruction counts per kernel.
usage: tools/kernel_resources.py [--isa] [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "raytracer-rs_amd", "csrc", "kernels.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize"]


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("mi355rt::", "")
    except OSError:
        return name


def main():
    args = sys.argv[1:]
    isa = "--isa" in args
    extra = [a for a in args if a != "--isa"]
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", SRC, "-o", "/tmp/kernels_res.o"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp")
    if isa and r.returncode == 0:
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["--save-temps", "-c", SRC, "-o", "/tmp/kernels_res2.o"], capture_output=True, text=True, cwd="/tmp")
    if r.returncode != 0:
        print(r.stderr)
        sys.exit(1)
    cur = None
    rows = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: +(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = demangle(t.split(":", 1)[1].strip())
            rows[cur] = {}
        elif cur and ":" in t:
            k, v = t.split(":", 1)
            rows[cur][k.strip()] = v.strip()
    print("%-44s %5s %5s %5s %8s %4s %8s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS"))
    for k, d in rows.items():
        print("%-44s %5s %5s %5s %8s %4s %8s" % (k[:44], d.get("VGPRs"), d.get("AGPRs"), d.get("TotalSGPRs"), d.get("ScratchSize [bytes/lane]"),
                                                 d.get("Occupancy [waves/SIMD]"), d.get("LDS Size [bytes/block]")))
    if isa:
        s = [f for f in os.listdir("/tmp") if f.startswith("kernels-hip-amdgcn") and f.endswith(".s")]
        if s:
            path = os.path.join("/tmp", s[0])
            text = open(path).read()
            os.replace(path, "/tmp/kernels.s")
            print("\nISA: /tmp/kernels.s")
            name = None
            cnt = {}
            for line in text.splitlines():
                m = re.match(r"^(_ZN\S+):", line)
                if m:
                    name = demangle(m.group(1)); cnt[name] = dict(valu=0, salu=0, vmem=0, lds=0, scratch=0)
                    continue
                t = line.strip()
                if not name or not t or t.startswith((";", ".")):
                    continue
                op = t.split()[0]
                if op.startswith("v_"):
                    cnt[name]["valu"] += 1
                elif op.startswith("s_"):
                    cnt[name]["salu"] += 1
                elif op.startswith(("global_", "buffer_", "flat_")):
                    cnt[name]["vmem"] += 1
                elif op.startswith("scratch_"):
                    cnt[name]["scratch"] += 1
                elif op.startswith("ds_"):
                    cnt[name]["lds"] += 1
            print("%-44s %6s %6s %6s %6s %7s" % ("kernel (static instruction counts)", "VALU", "SALU", "VMEM", "LDS", "scratch"))
            for k, c in cnt.items():
                if c["valu"]:
                    print("%-44s %6d %6d %6d %6d %7d" % (k[:44], c["valu"], c["salu"], c["vmem"], c["lds"], c["scratch"]))


if __name__ == "__main__":
    main()
