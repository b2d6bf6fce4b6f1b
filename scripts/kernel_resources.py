"""Static resource table of every gfx950 kernel in the product library: VGPRs, SGPRs, scratch bytes, static LDS,
register spills and the launch bound -- read from the code objects' AMDGPU metadata notes, no GPU needed.

    python scripts/kernel_resources.py [> profiles/rNN_kernel_resources.txt]

Each .hip source is compiled device-only with the product flags (nerfmeshes_amd/build.py) into a temporary
directory, the gfx950 code object is unbundled and `llvm-readelf --notes` is parsed.  What to look for: `scr` must be
0 for every fp32 MLP kernel (scratch stores go through to HBM), `v` decides waves per SIMD (512 / v, at most 8).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfmeshes_amd import build as B  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = (("v", "vgpr_count"), ("s", "sgpr_count"), ("scr", "private_segment_fixed_size"),
          ("lds", "group_segment_fixed_size"), ("spill_v", "vgpr_spill_count"), ("spill_s", "sgpr_spill_count"),
          ("wg", "max_flat_workgroup_size"))


def kernels_of(src, tmp):
    stem = os.path.splitext(src)[0]
    bundle, elf = os.path.join(tmp, stem + ".co"), os.path.join(tmp, stem + ".elf")
    subprocess.run([B.hipcc()] + B.FLAGS + ["--cuda-device-only", "-c", os.path.join(B.CSRC, src), "-o", bundle], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={bundle}", f"--output={elf}"], check=True)
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], capture_output=True, text=True, check=True).stdout
    rows = []
    for block in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        def field(key):
            m = re.search(r"\." + key + r":\s+(\S+)", block)
            return m.group(1) if m else "?"
        name = subprocess.run(["c++filt", field("name")], capture_output=True, text=True).stdout.strip()
        rows.append((re.sub(r"^void ", "", re.sub(r"\(.*", "", name)), [field(k) for _, k in FIELDS]))
    return rows


def main():
    print(f"{'source':16s} " + " ".join(f"{h:>7s}" for h, _ in FIELDS) + "  kernel")
    with tempfile.TemporaryDirectory() as tmp:
        for src in B.SOURCES:
            if not src.endswith(".hip"):
                continue
            for name, vals in sorted(kernels_of(src, tmp)):
                print(f"{src:16s} " + " ".join(f"{v:>7s}" for v in vals) + "  " + name)


if __name__ == "__main__":
    main()
