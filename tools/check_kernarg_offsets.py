#!/usr/bin/env python3
"""track_post_kernel reads some of its by-value argument structs straight from the kernel-argument segment (kernarg_late in
parc_amd/csrc/parc_kin.hip) at offsets it derives from the struct sizes.  This tool compiles the source to assembly and checks
those offsets against the argument layout the compiler recorded in the code object's metadata.  Exit code 0 = they agree."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def metadata_offsets(kernel="track_post_kernel"):
    from parc_amd import _hip
    src = os.path.join(ROOT, "parc_amd", "csrc", "parc_kin.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        hipcc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
        subprocess.check_call([hipcc, "--offload-arch=gfx950"] + _hip.OPT_LEVEL.get("parc_kin.hip", "-O3").split() +
                              ["-std=c++17", "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
        text = open(out).read()
    # amdhsa.kernels metadata: a list of kernels, each with .args (offset / size / value_kind) followed by .name
    blocks = text.split("  - .agpr_count:")
    for blk in blocks:
        m = re.search(r"\.name:\s+(\S*%s\S*)" % kernel, blk)
        if m:
            return [(int(o), int(s), k) for o, s, k in re.findall(r"\.offset:\s+(\d+)\s+\.size:\s+(\d+)\s+\.value_kind:\s+(\w+)", blk)]
    raise RuntimeError("kernel metadata not found")


def expected_offsets():
    from parc_amd import _hip
    structs = [_hip.CharModelS, _hip.MotionLibS, _hip.TerrainS, _hip.TrackCfgS, _hip.EnvBuffersS]
    offs, o = [], 0
    for s in structs:
        o = (o + 7) & ~7
        offs.append((o, ctypes.sizeof(s)))
        o += ctypes.sizeof(s)
    return offs


def main():
    meta = [a for a in metadata_offsets() if a[2] == "by_value"][:5]
    exp = expected_offsets()
    ok = [(o, s) for o, s, _ in meta] == exp
    print("metadata:", [(o, s) for o, s, _ in meta])
    print("expected:", exp)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
