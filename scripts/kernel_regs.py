"""Register / LDS / spill table of the kernels in an AMDGPU .s file (hipcc --save-temps): python scripts/kernel_regs.py file.s"""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    try: name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception: pass
    name = re.sub(r"\(.*", "", name).replace("avmoe::(anonymous namespace)::", "").replace("avmoe::", "")
    print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>3s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>3s} lds {g('group_segment_fixed_size'):>6s}")
