#!/usr/bin/env python3
"""VGPR / SGPR / spill / LDS / scratch figures of every kernel in a built object or shared library, read from the
code-object notes (what the hardware allocates; rocprofv3's database reports something else in its vgpr column).
usage: kernel_resources.py file.o|file.so [name-filter]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def code_objects(path):
    """gfx950 code objects embedded in `path` (.hip_fatbin: one clang offload bundle per translation unit)."""
    d = tempfile.mkdtemp()
    fb = os.path.join(d, "fatbin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, path])
    blob = open(fb, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    out = []
    for i, s in enumerate(starts):
        e = starts[i + 1] if i + 1 < len(starts) else len(blob)
        part = os.path.join(d, "bundle%d" % i)
        open(part, "wb").write(blob[s:e])
        co = os.path.join(d, "co%d" % i)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co):
            out.append(co)
    return out


def kernels(co):
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    cur = {}
    for line in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):
            yield cur
            cur = {}
        if k in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "agpr_count",
                 "group_segment_fixed_size", "private_segment_fixed_size", "max_flat_workgroup_size"):
            cur[k] = v
        if k == "wavefront_size" and cur.get("name"):
            yield cur
            cur = {}


def demangle(n):
    try:
        r = subprocess.run(["c++filt", n], capture_output=True, text=True)
        return r.stdout.strip() or n
    except OSError:
        return n


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    print("%-70s %5s %5s %7s %7s %7s %8s" % ("kernel", "vgpr", "sgpr", "v_spill", "s_spill", "lds", "scratch"))
    seen = set()
    for co in code_objects(path):
        for k in kernels(co):
            name = demangle(k["name"]).replace("(anonymous namespace)::", "")
            name = re.sub(r"\(.*\)$", "", name).replace("void ", "")
            if flt not in name or name in seen:
                continue
            seen.add(name)
            print("%-70s %5s %5s %7s %7s %7s %8s" % (name[:70], k.get("vgpr_count"), k.get("sgpr_count"),
                                                       k.get("vgpr_spill_count"), k.get("sgpr_spill_count"),
                                                       k.get("group_segment_fixed_size"), k.get("private_segment_fixed_size")))


if __name__ == "__main__":
    main()
