#!/usr/bin/env python
"""Condense the log of a survey run of the GPU suite into profiles/parity_survey_<tag>.json:
    EG_PARITY_RECORD=$PWD/gpurun_out/parity_record.jsonl python -m pytest tests -m gpu -q      (on the GPU box)
    python tools/parity_survey.py gpurun_out/parity_record.jsonl profiles/parity_survey_r02.json
Every three-way comparison of tests/parity.py is one line: distance of the backend and of the oracle from the
float64 shadow, relative to max|exact|."""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = [json.loads(line) for line in open(src) if line.strip()]
TOL = 1e-5
short = lambda t: t.split("::")[-1].split(" ")[0].split("[")[0]
above = [r for r in rows if r["e_gpu"] > TOL]
out = {
    "comparisons": len(rows),
    "backend_above_1e-5": len(above),
    "oracle_above_1e-5": sum(1 for r in rows if r["e_ref"] > TOL),
    "backend_max": max((r["e_gpu"] for r in rows), default=0.0),
    "oracle_max": max((r["e_ref"] for r in rows), default=0.0),
    "backend_above_1e-5_cases": [
        {"test": short(r["test"]), "what": r["what"], "backend_vs_exact": r["e_gpu"], "oracle_vs_exact": r["e_ref"],
         "longest_reduction": r["n"]} for r in sorted(above, key=lambda r: -r["e_gpu"])],
    # (tensors of fewer than 64 elements are held to the cap only, tests/parity.py: their lines carry "(cap only)"
    #  and compare the shadow with itself on the oracle side)
    "backend_farther_than_twice_the_oracle_and_above_1e-5": sum(1 for r in above if r["e_gpu"] > 2 * r["e_ref"] and "(cap only)" not in r["what"]),
    "cap_only_comparisons_above_1e-5": sum(1 for r in above if "(cap only)" in r["what"]),
    "note": "distances relative to max|exact| of the compared tensor — for gradients of fewer than 64 elements relative, element by "
            "element, to the sum of the magnitudes of the summed terms (lines marked 'vs sum of |terms|'); the rule of "
            "tests/parity.py: backend <= max(1e-5, min(2 x oracle, 1e-3)).  No cap-only comparisons since round 3.",
}
json.dump(out, open(dst, "w"), indent=1)
print({k: v for k, v in out.items() if not isinstance(v, list)})
