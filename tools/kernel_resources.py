"""Register / scratch use of every kernel in the built product library, read from the code objects' metadata notes (no GPU):
the gfx950 code objects are unbundled from the library's .hip_fatbin section and `llvm-readelf --notes` prints their
amdhsa.kernels records. Used by tests/test_abi_and_host.py (no kernel may spill outside an explicit allow-list) and as a
script: `python tools/kernel_resources.py [lib.so]` prints name, VGPRs, AGPRs, SGPR / VGPR spills, scratch bytes, LDS bytes."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path):
    """the gfx950 ELF images bundled into a host library (one bundle per translation unit)"""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib_path, os.path.join(tmp, "copy.so")], check=True)
        blob = open(fat, "rb").read()
    out = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        (n,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "amdgcn" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(MAGIC, pos + len(MAGIC))
    return out


FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")


def kernels(lib_path=None):
    """{demangled kernel name: {vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size, ...}}"""
    lib_path = lib_path or os.path.join(ROOT, "large-steps-pytorch_amd", "lib", "liblargesteps_hip.so")
    res = {}
    for image in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as fh:
            fh.write(image)
            fh.flush()
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", fh.name], check=True, capture_output=True, text=True).stdout
        # records of amdhsa.kernels: "- .agpr_count: 0" starts one, the fields follow in alphabetical order
        for rec in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
            rec = ".agpr_count:" + rec
            name = re.search(r"\.name:\s+(\S+)", rec).group(1).strip("'\"")
            vals = {f: int(m.group(1)) for f in FIELDS for m in [re.search(rf"\.{f}:\s+(\d+)", rec)] if m}
            res[name] = vals
    names = list(res)
    import shutil
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    dem = subprocess.run([filt] + names, capture_output=True, text=True).stdout.splitlines() if names and filt else []
    if len(dem) == len(names):
        res = {re.sub(r"\s*\(.*$", "", d): res[n] for d, n in zip(dem, names)}
    return res


if __name__ == "__main__":
    ks = kernels(sys.argv[1] if len(sys.argv) > 1 else None)
    print(f"{'kernel':70s} vgpr agpr sgpr  vspill sspill scratch    lds")
    for name in sorted(ks):
        k = ks[name]
        print(f"{name[:70]:70s} {k.get('vgpr_count', 0):4d} {k.get('agpr_count', 0):4d} {k.get('sgpr_count', 0):4d} {k.get('vgpr_spill_count', 0):7d} "
              f"{k.get('sgpr_spill_count', 0):6d} {k.get('private_segment_fixed_size', 0):7d} {k.get('group_segment_fixed_size', 0):6d}")
    bad = {n: k for n, k in ks.items() if k.get("vgpr_spill_count", 0) or k.get("private_segment_fixed_size", 0)}
    print(f"{len(ks)} kernels, {len(bad)} with VGPR spills or scratch")
