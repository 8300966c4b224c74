"""Fingerprint of the kernel sources (csrc/*.hip, *.h): profiles/traffic.json and valu_counts.json carry the fingerprint of the
code they were collected on, bench.py compares it with the code it runs and says so when a table is stale."""
import hashlib
import os


def kernel_source_hash():
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha1()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]
