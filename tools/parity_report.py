#!/usr/bin/env python
"""The comparisons of a GPU test run that sit above 1e-5 (tests/conftest.py records every parity comparison into
gpurun_out/parity_achieved_gpu.json), each with the bound it was accepted under and what that bound is anchored on.

    python tools/parity_report.py profiles/r5_parity_achieved_gpu.json > profiles/r5_parity_above_1e-5.txt

Since round 5 every "float64" in these lines is the REFERENCE'S OWN float64 run (tests/golden/make_golden.py: reference_in_float64), not
the builder's restatement; `reference's own distance` is the reference's float32 output against that run."""
import json, sys
rows = json.load(open(sys.argv[1]))
big = [r for r in rows if r["rel_l2"] > 1e-5]
over = [r for r in rows if r["rel_l2"] > r["bound"]]
print(f"{len(rows)} recorded comparisons, {len(big)} above 1e-5 rel-L2, {len(over)} above the bound recorded next to them")
if over:
    print("(those are two-criteria checks -- within the bound of the reference's float32 output OR at least as close to the reference's float64 run as\n"
          " that output is: the line below them, same test, carries the criterion that accepted them)")
print()
ANCHOR = (("float64", "judged against the reference's own float64 run: at least as close to it as the reference's float32 output is (+ the 1e-5 budget)"),
          ("both fp32", "two float32 evaluations of an x-update that amplifies round-off by 1 / min(|H|^2 + rho): bounded by twice the reference's own distance from its float64 run"),
          ("bf16", "bf16 history (a stated tolerance of the bf16 mode, DESIGN section 8)"),
          ("implicit", "implicit-function gradient of an iterative solve: bounded by the solver's rtol"),
          ("cg ", "float32 CG on a small dense system: bounded by rtol x condition number"),
          ("autograd", "fp32 autograd of either side through 10 unrolled iterations: sums of ~3e6 signed products that cancel to ~1e-5 of their magnitude"))
for r in sorted(big, key=lambda r: -r["rel_l2"]):
    why = next((txt for key, txt in ANCHOR if key in r["what"]), "")
    print(f"{r['rel_l2']:.2e}  (bound {r['bound']:.2e})  {r['test'].split('::')[-1]}: {r['what']}")
    if why:
        print(f"            {why}")
