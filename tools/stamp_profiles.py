#!/usr/bin/env python3
"""Copies files of a collection (gpurun_out/<tag>/, tools/collect_round5.sh) into profiles/ with the collection's stamp.json written
INTO each of them: JSON objects / lines get a "stamp" key (JSON arrays are wrapped: {"stamp": ..., "records": [...]} is avoided --
consumers index them -- so a first element {"stamp": ...} is prepended), CSV and text files a leading `# stamp: {...}` line.
    python tools/stamp_profiles.py gpurun_out/r4x  bench.json=r04_bench_n1.json  trace/run_kernel_stats.csv=r04_kernel_stats.csv ..."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    stamp = json.load(open(os.path.join(src, "stamp.json")))
    stamp["collection"] = src
    for pair in sys.argv[2:]:
        a, b = pair.split("=")
        text = open(os.path.join(src, a)).read()
        dst = os.path.join(ROOT, "profiles", b)
        if b.endswith(".json"):
            try:
                obj = json.loads(text)
            except ValueError:                              # a bench log: the JSON line is the last line that parses
                obj = json.loads([l for l in text.splitlines() if l.startswith("{")][-1])
            if isinstance(obj, dict):
                obj["stamp"] = stamp
            else:
                obj = [{"stamp": stamp}] + list(obj)
            out = json.dumps(obj, indent=1) + "\n"
        elif b.endswith(".jsonl"):
            lines = []
            for l in text.splitlines():
                if l.startswith("{"):
                    o = json.loads(l)
                    o["stamp"] = stamp
                    lines.append(json.dumps(o))
            out = "\n".join(lines) + "\n"
        else:
            out = "# stamp: " + json.dumps(stamp) + "\n" + text
        with open(dst, "w") as f:
            f.write(out)
        print("profiles/" + b)


if __name__ == "__main__":
    main()
