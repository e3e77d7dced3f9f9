#!/usr/bin/env python
"""Condense the log of a survey run of the GPU suite into profiles/parity_survey_<tag>.json:
    EG_PARITY_RECORD=$PWD/gpurun_out/parity_record.jsonl python -m pytest tests -m gpu -q      (on the GPU box)
    python tools/parity_survey.py gpurun_out/parity_record.jsonl profiles/parity_survey_r06.json
Every three-way comparison of tests/parity.py is one line: distance of the backend and of the oracle from the
float64 shadow (e_gpu, e_ref; relative to max|exact|) and — round 5, the comparison in BASELINE.json's own words
("outputs match the reference LLVM CPU path on identical inputs within 1e-5 relative") — the distance of the backend
from the ORACLE directly (e_go, relative to max|oracle|).  Two-way comparisons made with conftest.rel_err are logged
too ("direct") with their label; in the full-size tests of BASELINE configs 1, 2 and 4 the second operand is the oracle, the
XOR test (config 3) labels each of its comparisons (backend vs exact, oracle vs exact, backend vs oracle).

The reference's summation order is passes.nim:700-745 (sequential float32 accumulation in increasing index order of
the reduction loops); the backend sums in MFMA tiles and trees.  Where the two float32 results differ by more than
1e-5, the file lists the comparison with the oracle's OWN distance from the exact value next to it: that distance is
what excuses the exception (the reference's result is that far from the value both compute)."""
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = [json.loads(line) for line in open(src) if line.strip()]
TOL = 1e-5
short = lambda t: t.split("::")[-1].split(" ")[0].split("[")[0]
three = [r for r in rows if "e_gpu" in r]
direct = [r for r in rows if "direct" in r]
above = [r for r in three if r["e_gpu"] > TOL]

CONFIGS = [
    ("configs[0] matmul 256^3 (reference CPU case)", r"cfg1"),
    ("configs[1] matmul 4096^3", r"cfg2"),
    ("configs[2] xor_from_scratch, batch 65536", r"cfg3"),
    ("configs[3] conv2 3x3 256x256x64->64", r"cfg4"),
    ("configs[4] dense 784-512-10 step, 65536-sample shard", r"cfg5"),
]


def config_of(test):
    for name, pat in CONFIGS:
        if re.search(pat, test):
            return name
    return None


per_config = {}
for name, _ in CONFIGS:
    t3 = [r for r in three if config_of(short(r["test"])) == name]
    d2 = [r for r in direct if config_of(short(r["test"])) == name]
    ex = [r for r in t3 if r.get("e_go", 0.0) > TOL]
    per_config[name] = {
        "three_way_comparisons": len(t3),
        "backend_vs_oracle_max": max((r.get("e_go", 0.0) for r in t3), default=None),
        "backend_vs_exact_max": max((r["e_gpu"] for r in t3), default=None),
        "oracle_vs_exact_max": max((r["e_ref"] for r in t3), default=None),
        "direct_comparisons": [{"what": r["what"], "distance": r["direct"]} for r in d2],
        "direct_backend_max": max((r["direct"] for r in d2 if "ORACLE vs exact" not in r["what"]), default=None),
        "backend_vs_oracle_above_1e-5": [
            {"test": short(r["test"]), "what": r["what"], "backend_vs_oracle": r["e_go"], "oracle_vs_exact": r["e_ref"],
             "backend_vs_exact": r["e_gpu"], "longest_reduction": r["n"],
             "excused_by": ("the oracle's own float32 summation is %.1e from the exact value of the same step"
                            % r["e_ref"]) if r["e_ref"] > 0.5 * r["e_go"] else "NOT excused by the oracle's distance"}
            for r in sorted(ex, key=lambda r: -r["e_go"])],
    }

go_above = [r for r in three if r.get("e_go", 0.0) > TOL]
out = {
    "comparisons": len(three),
    "backend_above_1e-5": len(above),
    "oracle_above_1e-5": sum(1 for r in three if r["e_ref"] > TOL),
    "backend_max": max((r["e_gpu"] for r in three), default=0.0),
    "oracle_max": max((r["e_ref"] for r in three), default=0.0),
    "backend_vs_oracle_above_1e-5_all_tests": len(go_above),
    "backend_vs_oracle_max_all_tests": max((r.get("e_go", 0.0) for r in three), default=0.0),
    "backend_vs_oracle_above_1e-5_not_excused": sum(1 for r in go_above if not r["e_ref"] > 0.5 * r["e_go"]),
    "per_baseline_config": per_config,
    "backend_vs_oracle_above_1e-5_cases_outside_the_configs": [
        {"test": short(r["test"]), "what": r["what"], "backend_vs_oracle": r["e_go"], "oracle_vs_exact": r["e_ref"],
         "backend_vs_exact": r["e_gpu"], "longest_reduction": r["n"]}
        for r in sorted(go_above, key=lambda r: -r["e_go"]) if config_of(short(r["test"])) is None],
    "backend_above_1e-5_cases": [
        {"test": short(r["test"]), "what": r["what"], "backend_vs_exact": r["e_gpu"], "oracle_vs_exact": r["e_ref"],
         "backend_vs_oracle": r.get("e_go"), "longest_reduction": r["n"]} for r in sorted(above, key=lambda r: -r["e_gpu"])],
    "backend_farther_than_twice_the_oracle_and_above_1e-5": sum(1 for r in above if r["e_gpu"] > 2 * r["e_ref"] and "(cap only)" not in r["what"]),
    "direct_two_way_comparisons": len(direct),
    "direct_two_way_max": max((r["direct"] for r in direct), default=0.0),
    "note": "e_gpu / e_ref: distances from the float64 shadow relative to max|exact| of the compared tensor (gradients of fewer "
            "than 64 elements: element by element relative to the sum of the magnitudes of the summed terms, lines marked "
            "'vs sum of |terms|').  backend_vs_oracle: max|backend - oracle| / max|oracle| (small gradients: same scale as the "
            "other two).  An exception is 'excused' when the oracle itself is at least half as far from the exact value as "
            "the backend is from the oracle.  The rule the suite asserts (tests/parity.py): backend <= max(1e-5, min(2 x oracle, 1e-3)) "
            "against the exact value.",
}
json.dump(out, open(dst, "w"), indent=1)
print({k: v for k, v in out.items() if not isinstance(v, (list, dict))})
for name, c in per_config.items():
    print(name, {k: v for k, v in c.items() if not isinstance(v, list)}, "exceptions:", len(c["backend_vs_oracle_above_1e-5"]))
