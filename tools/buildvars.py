#!/usr/bin/env python3
"""Build tuning variants of the library in parallel:  tools/buildvars.py "name=-DFOO=1;-DBAR=2" name2=...
Each variant becomes ni-slam_amd/libnislam_kcc_hip_<name>.so (benchmark them with tools/runvar.sh _<name> ...);
the default library is rebuilt too.  Exits non-zero if any build fails."""
import concurrent.futures as cf, importlib.util, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("b", os.path.join(root, "ni-slam_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
V = {"": []}
for a in sys.argv[1:]:
    k, _, d = a.partition("=")
    V[k] = [x for x in d.split(";") if x]          # defs separated by ";" (macro values may contain commas)
def go(kv):
    k, d = kv
    try:
        return k, b.build(defs=d, suffix="_" + k if k else ""), True
    except Exception as e:
        return k, str(e)[-400:], False
ok = True
with cf.ThreadPoolExecutor(8) as ex:
    for k, r, good in ex.map(go, V.items()):
        print(k or "base", r if not good else os.path.basename(r)); ok &= good
sys.exit(0 if ok else 1)
