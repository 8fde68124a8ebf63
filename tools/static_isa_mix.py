"""Static instruction mix of the HIP kernels from their gfx950 assembly (no GPU needed): compiles lt-mapper_amd/csrc/ltm_k_projection.hip with the
Makefile's flags and --cuda-device-only -S, then counts, per kernel whose mangled name contains a given substring, the VALU / SALU / LDS / global
instructions, the loop headers, the register and LDS budget, and the most frequent VALU mnemonics.  How profiles/r4_vote_kernel_static_instruction_mix.txt
was made (the per-point figures there come from reading one unrolled block of the listing by hand).

  python tools/static_isa_mix.py k_vote_map_cullILb1ELb1 k_map_rimg_blockminILb1ELb1
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "lt-mapper_amd", "csrc"), "--cuda-device-only", "-S"]


def main():
    wanted = sys.argv[1:] or ["k_vote_map_cullILb1ELb1", "k_map_rimg_blockminILb1ELb1"]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(ROOT, "lt-mapper_amd", "csrc", "ltm_k_projection.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN3ltm\w+:", l)]
    for w in wanted:
        for i, name in starts:
            if w not in name:
                continue
            end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
            body = [l.split()[0] for l in lines[i:end] if l.strip() and not l.strip().startswith(";") and not l.startswith((".", "_")) and l.split()]
            valu = [b for b in body if b.startswith("v_")]
            meta = {}
            for l in lines[end:end + 400]:
                m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size)\s+(\d+)", l)
                if m:
                    meta[m.group(1)] = int(m.group(2))
                if "end_amdhsa_kernel" in l:
                    break
            mix = collections.Counter(re.sub(r"_e(32|64)$", "", v) for v in valu).most_common(12)
            print(f"{name[:90]}\n  VALU {len(valu)}  SALU {sum(b.startswith('s_') for b in body)}  LDS {sum(b.startswith('ds_') for b in body)}  "
                  f"global {sum(b.startswith('global_') for b in body)}  loops {sum('Loop Header' in l for l in lines[i:end])}  {meta}\n  {mix}")


if __name__ == "__main__":
    main()
