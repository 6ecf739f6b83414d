"""profiles/pmc_latest.json holds per-env-step HBM traffic and instruction counts that bench.py scales to its own launches
(`roofline.traffic`, `valu`): stored figures, not live ones.  They describe the shipped kernels only while the kernel sources have
not changed since the profile run -- this test enforces that: every record carries the hash of the device sources it was measured on
(tools/src_hash.py: comments and white space removed), and it must equal the hash of the tree."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_pmc_records_describe_the_kernels_in_the_tree():
    from src_hash import source_hash
    d = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_latest.json')))
    want = source_hash()
    recs = [(k, v) for k, v in d.items() if isinstance(v, dict) and 'shape' in v]
    for m in d.get('more', []):
        recs += [(k, v) for k, v in m.items() if isinstance(v, dict)]
    assert recs, 'no PMC records'
    stale = sorted({f"{k} {v.get('shape')}" for k, v in recs if v.get('src_hash') != want})
    assert not stale, ('profiles/pmc_latest.json was measured on other kernel sources than csrc/ holds now (re-run tools/gpu_profile_all.sh + '
                       f'tools/collect_profiles.sh, or revert the kernel change): {stale[:3]} ... tree hash {want}')


def test_comment_only_edits_keep_the_hash(tmp_path):
    from src_hash import strip
    a = 'int f(int x) { return x + 1; }  // adds one\n/* block */\n'
    b = 'int f(int x) {\n  return x + 1;   // something else entirely\n}\n'
    assert strip(a) == strip(b) and strip(a) != strip(a.replace('+ 1', '+ 2'))
