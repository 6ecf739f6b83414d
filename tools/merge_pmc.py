#!/usr/bin/env python3
"""merge_pmc.py main.json other.json ... > profiles/pmc_latest.json : the PMC records of the default workload (first file, kept at
the top level: k_closed / k_stages / raycast_stage) with those of further launch shapes (BASELINE configs 3, 4, 5 and the 65 536-env
shapes of config 2) under `more`, the list bench.py searches for the shape of its own launch."""
import json
import sys

main = json.load(open(sys.argv[1]))
main['more'] = []
for f in sys.argv[2:]:
    d = json.load(open(f))
    main['more'].append({k: v for k, v in d.items() if isinstance(v, dict) and 'shape' in v})
hashes = {main.get('src_hash')} | {v.get('src_hash') for m in main['more'] for v in m.values()}
if len(hashes) != 1:
    sys.exit(f'merge_pmc.py: the records were measured on different kernel sources: {sorted(map(str, hashes))}')
json.dump(main, sys.stdout, indent=1)
