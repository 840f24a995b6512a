#!/usr/bin/env python3
"""profiles/k1_traffic.json from the PMC summaries tools/profile_round.sh writes: per INSTANTIATION of nn_fast_kernel
(keyed by its full template string, e.g. "nn_fast_kernel<10, 16, 1728, false, 8>") the HBM bytes per launch
(pmc_hbm_traffic.txt: FETCH_SIZE / WRITE_SIZE, with the gfx950 correction the MI355X_MICROARCH.md HBM section
prescribes: FETCH_SIZE counts wide coalesced reads at half size) and, when the instruction-mix passes ran
(pmc_instruction_mix.txt), VALU / LDS instructions, waves and LDS bank-conflict cycles per launch.  bench.py quotes an
entry only for the instantiation its timed windows really ran.
usage: make_k1_traffic.py <pmc_hbm_traffic.txt> <out.json> <tag> [pmc_instruction_mix.txt]"""
import json
import re
import sys

CORRECTION = ("gfx950: FETCH_SIZE counts 128-B fabric requests at 64 B for wide coalesced reads -> x2 "
              "(MI355X_MICROARCH.md HBM section); WRITE_SIZE taken as is; Infinity-Cache hits are included in both")


def blocks(path):
    """{kernel head line: {counter: (mean, n)}} of one pmc_summary.py output"""
    out = {}
    for b in re.split(r"\n(?=\S)", open(path).read()):
        lines = b.splitlines()
        if not lines:
            continue
        out[lines[0].strip()] = {m.group(1): (float(m.group(2)), int(m.group(3)))
                                 for m in re.finditer(r"(\w+)\s+([0-9.]+)\s+\(n=(\d+)\)", b)}
    return out


def main(summary, out, tag, mix=None):
    entries = {}
    for head, vals in blocks(summary).items():
        if not head.startswith("nn_fast_kernel<") or "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
            continue
        fetch_kb, n = vals["FETCH_SIZE"]
        write_kb, _ = vals["WRITE_SIZE"]
        entries[head] = {"dispatches": n, "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
                         "traffic_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0}
    if not entries:
        raise SystemExit("no nn_fast_kernel<...> block in " + summary)
    if mix:
        for head, vals in blocks(mix).items():
            if head in entries and "SQ_INSTS_VALU" in vals and "SQ_WAVES" in vals:
                e = entries[head]
                e["waves"] = vals["SQ_WAVES"][0]
                e["valu_insts"] = vals["SQ_INSTS_VALU"][0]
                e["valu_per_wave"] = vals["SQ_INSTS_VALU"][0] / max(vals["SQ_WAVES"][0], 1.0)
                for k_in, k_out in (("SQ_INSTS_LDS", "lds_insts"), ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict_cycles"),
                                    ("SQ_LDS_IDX_ACTIVE", "lds_active_cycles"), ("SQ_THREAD_CYCLES_VALU", "valu_thread_cycles"),
                                    ("SQ_INSTS_VALU_INT32", "valu_int32_insts")):
                    if k_in in vals:
                        e[k_out] = vals[k_in][0]
    doc = {"source": f"rocprofv3 --pmc passes over tools/exp_align.py (one pass per counter group, --kernel-trace only), MI355X, "
                     f"profiles/{tag}_pmc_hbm_traffic.txt" + (f" + profiles/{tag}_pmc_instruction_mix.txt" if mix else "")
                     + "; tools/profile_round.sh",
           "correction": CORRECTION, "entries": entries}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rXX", sys.argv[4] if len(sys.argv) > 4 else None)
