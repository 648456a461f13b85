#!/usr/bin/env python3
"""Digest of the PMC summaries (tools/pmc_summary.py output) into profiles/traffic.json, which bench.py
quotes as roofline.traffic:  tools/traffic_json.py <dir with <tag>_<workload>_pmc_summary.txt> <tag> > profiles/traffic.json"""
import json
import os
import re
import sys

NOTE = ("rocprofv3 --pmc passes of tools/pmc.sh (one run per counter group), mean per dispatch; hbm bytes = (2 x FETCH_SIZE + "
        "WRITE_SIZE) KB, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; valu_busy_fraction = SQ_ACTIVE_INST_VALU x 4 "
        "cycles / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs): the share of SIMD time a VALU instruction was executing; "
        "wave_wait_fraction = SQ_WAIT_ANY / SQ_WAVE_CYCLES; l1 = vector cache (TCP), l2 = TCC, 128-byte requests")


def sections(path):
    out, cur = {}, None
    for line in open(path):
        if line.startswith("== "):
            cur = out.setdefault(line[3:].strip(), {})
        else:
            m = re.match(r"\s+(\S+)\s+n=(\d+)\s+mean=(\S+)", line)
            if m and cur is not None:
                cur[m.group(1)] = float(m.group(3))
    return out


def main():
    d, tag = sys.argv[1], sys.argv[2]
    res = {}
    for w in ("c1", "c2", "c3", "c4", "c5", "mesh"):
        p = os.path.join(d, "%s_%s_pmc_summary.txt" % (tag, w))
        if not os.path.exists(p):
            continue
        sec = sections(p)
        # the dominant kernel: the timed (STATS = false) trace kernel with the most VALU instructions; the <true>
        # instantiation is bench.py's untimed counters launch
        names = [k for k in sec if "trace_" in k and ("<false>" in k or "<false," in k) and "SQ_INSTS_VALU" in sec[k]]
        k = max(names, key=lambda n: sec[n]["SQ_INSTS_VALU"])
        c = sec[k]
        short = re.sub(r"^.*(trace_\w+<[\w, ]+>).*$", r"\1", k)
        gui_per_xcd = c["GRBM_GUI_ACTIVE"] / 8.0
        res[w] = {
            "kernel": short,
            "fetch_size_kb": c["FETCH_SIZE"],
            "write_size_kb": c["WRITE_SIZE"],
            "hbm_bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
            "valu_wave_instructions": c["SQ_INSTS_VALU"],
            "valu_busy_fraction": c["SQ_ACTIVE_INST_VALU"] * 4 / (gui_per_xcd * 1024) if c.get("SQ_ACTIVE_INST_VALU") else None,
            "cycles_per_valu_instruction": c["SQ_ACTIVE_INST_VALU"] * 4 / c["SQ_INSTS_VALU"] if c.get("SQ_ACTIVE_INST_VALU") else None,
            "wave_wait_fraction": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
            "gpu_cycles": gui_per_xcd,
            "l2_hit": c.get("TCC_HIT_sum"), "l2_miss": c.get("TCC_MISS_sum"),
            "l1_accesses": c.get("TCP_TOTAL_CACHE_ACCESSES_sum"), "l1_to_l2_read_requests": c.get("TCP_TCC_READ_REQ_sum"),
            "l2_read_latency_cycles": (c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"]) if c.get("TCP_TCC_READ_REQ_sum") else None,
            "active_lanes_per_valu_instruction": c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"] if c.get("SQ_ACTIVE_INST_VALU") else None,
            "note": NOTE,
        }
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
