#!/usr/bin/env python
"""No kernel of the shipped library may use scratch memory (spilled registers).

Reads the per-kernel resource remarks hipcc prints with -Rpass-analysis=kernel-resource-usage -- __graft_entry__.build() keeps them next to every
object file (delta-prox_amd/build/*.o.resources.txt) -- and fails (exit status 1) if a kernel that is not a probe reports ScratchSize > 0.
Probe kernels (wrong results by design) would be exempt -- k_cols_p2<..., DBG != 0>, k_cols_probe_* -- none is instantiated any more.

    python tools/spill_check.py [-v]        (tools/spill_check.sh is the same call; tests/test_host_logic.py runs it)
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "delta-prox_amd", "build")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def is_probe(name):
    if "k_cols_probe" in name:
        return True
    m = re.match(r".*k_cols_p2<(\d+), (\d+), (\d+), (\d+), (\d+)>", name)
    return bool(m) and int(m.group(5)) != 0


def kernels():
    """[(source, kernel, vgprs, scratch bytes per lane, spilled vgprs)] of every kernel of the library"""
    files = sorted(glob.glob(os.path.join(BUILD, "*.o.resources.txt")))
    srcs = sorted(glob.glob(os.path.join(ROOT, "delta-prox_amd", "csrc", "*.hip")))
    if len(files) < len(srcs):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as ge
        ge.build()
        files = sorted(glob.glob(os.path.join(BUILD, "*.o.resources.txt")))
    rows = []
    for f in files:
        txt = open(f).read()
        for blk in txt.split("Function Name: ")[1:]:
            name = blk.split()[0]
            get = lambda key: int(re.search(key + r": (\d+)", blk).group(1))
            rows.append((os.path.basename(f).split(".o.")[0], name, get(r"VGPRs"), get(r"ScratchSize \[bytes/lane\]"), get(r"VGPRs Spill")))
    names = demangle([r[1] for r in rows])
    return [(src, names[n].replace("dpx::", ""), v, s, sp) for src, n, v, s, sp in rows]


def main(verbose=False):
    rows = kernels()
    bad = [r for r in rows if r[3] > 0 and not is_probe(r[1])]
    probes = [r for r in rows if r[3] > 0 and is_probe(r[1])]
    if verbose:
        for src, n, v, s, sp in rows:
            print(f"{src:22s} vgprs {v:3d} scratch {s:4d}  {n.split('(')[0][:110]}")
    print(f"spill_check: {len(rows)} kernels, {len(bad)} with scratch memory ({len(probes)} probe kernels with scratch are exempt)")
    for src, n, v, s, sp in bad:
        print(f"  SCRATCH {s} bytes / lane ({sp} spilled registers): {src}: {n.split('(')[0]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main("-v" in sys.argv))
