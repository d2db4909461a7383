#!/usr/bin/env python3
"""tests/tolerances.json from the observed errors of a full `pytest -m gpu` run on the GPU box.

Every bf16 / fp8 comparison in tests/ goes through gpu_util.observe(name, value, inline_tol), which appends the measured
error to gpurun_out/observed_errors.jsonl.  This script turns that log into the table the same function then enforces:
tol = min(inline tolerance, max(2 x largest observed value, 1e-5)) -- so no tolerance in the suite has more than 2x
slack over what was measured, and DESIGN.md section 6 can quote the observed column.

    python tools/update_tolerances.py [gpurun_out/observed_errors.jsonl ...] > summary
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    paths = sys.argv[1:] or [os.path.join(ROOT, "gpurun_out", "observed_errors.jsonl")]
    out_path = os.path.join(ROOT, "tests", "tolerances.json")
    table = {}
    if os.path.exists(out_path):
        with open(out_path) as f:
            table = json.load(f)
    seen = {}
    for p in paths:
        with open(p) as f:
            for line in f:
                r = json.loads(line)
                e = seen.setdefault(r["name"], {"observed": 0.0, "inline": r["inline_tol"] if "inline_tol" in r else r["tol"], "kind": r["kind"]})
                e["observed"] = max(e["observed"], r["observed"])
                e["inline"] = max(e["inline"], r.get("inline_tol", r["tol"]))
    for name, e in seen.items():
        tol = min(e["inline"], max(2.0 * e["observed"], 1e-5))
        table[name] = {"observed": float(f"{e['observed']:.3e}"), "tol": float(f"{tol:.3e}"), "kind": e["kind"]}
    with open(out_path, "w") as f:
        json.dump(dict(sorted(table.items())), f, indent=0)
    print(f"{len(seen)} names updated, {len(table)} in {out_path}")


if __name__ == "__main__":
    main()
