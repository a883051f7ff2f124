import sys, importlib.util, os
spec = importlib.util.spec_from_file_location("build", "/root/repo/meshgraphnets.jl_amd/build.py")
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
name = sys.argv[1]; flags = [f for f in sys.argv[2:] if not f.startswith("split:")]
pf = [f[6:] for f in sys.argv[2:] if f.startswith("split:")]
print(b.build_variant(name, flags, per_file={"split.hip": pf}))
