#!/usr/bin/env python
"""sha256 (first 16 hex digits) over the kernel sources magical_amd/csrc/* in name order: the stamp that ties a committed counter
profile (profiles/rNN_pmc_*.json, written by tools/pmc_summary.py / pmc_alu_summary.py) to the sources it was measured on.
bench.py quotes such a profile only when its stamp equals the hash of the sources the library in use was built from."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16(root=ROOT):
    d = os.path.join(root, 'magical_amd', 'csrc')
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith(('.hip', '.h', '.inc', '.cpp')):
            h.update(name.encode() + b'\0')
            h.update(open(os.path.join(d, name), 'rb').read())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(csrc_sha16())
