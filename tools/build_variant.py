import sys, importlib.util, os
"""python tools/build_variant.py name [flag ...] [split:flag ...] [train:flag ...] [kernels:flag ...]: a library variant under lib/variants/ (plain flags
apply to every source, prefixed ones to that .hip file only)"""
spec = importlib.util.spec_from_file_location("build", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "meshgraphnets.jl_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
name = sys.argv[1]
flags, per_file = [], {}
for f in sys.argv[2:]:
    for pre in ("split", "train", "kernels"):
        if f.startswith(pre + ":"):
            per_file.setdefault(pre + ".hip", []).append(f[len(pre) + 1:])
            break
    else:
        flags.append(f)
print(b.build_variant(name, flags, per_file=per_file))
