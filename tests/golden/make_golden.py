"""Prints the oracle's md5 for every synthetic anchor in anchors.json (sanity helper).

The expected values in anchors.json come from the reference itself;
this script never writes them."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib  # noqa: E402
from povu_amd import workloads  # noqa: E402

anchors = json.load(open(os.path.join(HERE, "anchors.json")))["md5"]
for key, want in anchors.items():
    if ":" not in key:
        continue
    name, arg = key.split(":")
    if name == "chain_of_bubbles":
        g = workloads.chain_of_bubbles(int(arg))
    else:
        d, t = arg.split("x")
        g = workloads.nested_towers(int(d), int(t))
    got = hashlib.md5(oracle_lib.decompose(g)[1].encode()).hexdigest()
    print(key, got, "OK" if got == want else "MISMATCH")
