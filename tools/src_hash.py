#!/usr/bin/env python3
"""src_hash.py : a hash of the device sources (csrc/*.hip, csrc/*.h, csrc/build.sh, include/d2d.h) with comments and white space
removed -- the identity of the kernels a profile was taken on.  tools/summarize_profile.py stores it in every PMC record;
tests/test_profiles.py fails when profiles/pmc_latest.json (the per-env-step traffic / instruction figures bench.py scales to its
own launches) was produced from other kernel sources than the tree holds.  Comment-only edits do not change it."""
import glob
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip(text, hash_comments=False):
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    text = re.sub(r'//[^\n]*', ' ', text)
    if hash_comments:
        text = re.sub(r'(?m)^\s*#[^\n]*', ' ', text)
    return re.sub(r'\s+', ' ', text).strip()


def source_hash(root=ROOT):
    csrc = os.path.join(root, 'gym-drone2d-activeperception_amd', 'csrc')
    files = sorted(glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.h'))) + [os.path.join(root, 'include', 'd2d.h')]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b'\0' + strip(open(f).read()).encode() + b'\0')
    h.update(b'build.sh\0' + strip(open(os.path.join(csrc, 'build.sh')).read(), hash_comments=True).encode())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(source_hash())
