"""Extracts the known answers the reference's own conformance suite holds for the decompose path into
tests/golden/reference_vectors.json (DATA: inputs and expected outputs; no source text is copied).

Run in the build container, where /root/reference exists:   python tests/golden/extract_reference_vectors.py

Sources (read as text):
* tests/lean4_conformance/lean_reference.lean -- `fixtureStructureOutput?` (:953-976) names the expected
  structure of every fixture: its segments and links (the input graph), `flubbleBoundary` rows (boundary candidates,
  emission order) and `dummyNode` / `flubbleNode` rows (the PVST: order, id, endpoints, parent, children, depth).
* tests/lean4_conformance/src/main.rs `cycle_oracle_for_fixture` (:1594-1622) -- the cycle-equivalence partition of
  the black tree edges, by tree-edge id.
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
lean = open(os.path.join(REF, "tests/lean4_conformance/lean_reference.lean")).read()
rs = open(os.path.join(REF, "tests/lean4_conformance/src/main.rs")).read()


def definition(name):
    """Body of `def <name> ... :=` up to the next top-level `def` / `example` / `theorem`."""
    m = re.search(r"^def " + re.escape(name) + r"\b[^\n]*?:=", lean, re.M | re.S)
    if not m:
        m = re.search(r"^def " + re.escape(name) + r"\b.*?:=", lean, re.M | re.S)
    if not m:
        raise KeyError(name)
    rest = lean[m.end():]
    stop = re.search(r"^(def|example|theorem|/--|end|namespace)\b", rest, re.M)
    return rest[:stop.start()] if stop else rest


def expand(body, depth=0):
    """Inline the helper definitions a structure refers to by name (segments0123, minimalLinks, ...)."""
    if depth > 4:
        return body
    out = body
    for ident in set(re.findall(r"\b([a-z][A-Za-z0-9]*(?:Segments|Links)[A-Za-z0-9]*|segments0123)\b", body)):
        try:
            out = out.replace(ident, expand(definition(ident), depth + 1))
        except KeyError:
            pass
    return out


def strings(s):
    return re.findall(r'"([^"]*)"', s)


def parse_structure(body):
    body = expand(body)
    gfa = re.search(r'structureText\s+"([^"]+)"', body).group(1)
    segs = [(int(a), int(b), c) for a, b, c in re.findall(r'segmentJson\s+(\d+)\s+(\d+)\s+"([^"]*)"', body)]
    links = [(int(i), int(a), sa, int(b), sb) for i, a, sa, b, sb in
             re.findall(r'linkJson\s+(\d+)\s+(\d+)\s+"([+-])"\s+(\d+)\s+"([+-])"', body)]
    bounds = [dict(order=int(o), node_id=n, start=a, stop=z) for o, n, a, z in
              re.findall(r'flubbleBoundary\s+(\d+)\s+"([^"]+)"\s+"([^"]+)"\s+"([^"]+)"', body)]
    nodes = []
    for m in re.finditer(r'dummyNode\s+\[([^\]]*)\]', body):
        nodes.append(dict(order=0, node_id="1:0", kind="dummy", start=None, stop=None, parent=None, children=strings(m.group(1)), depth=0))
    for m in re.finditer(r'flubbleNode\s+(\d+)\s+(\d+)\s+"([^"]+)"\s+"([^"]+)"\s+"([^"]+)"\s+\((?:some\s+"([^"]+)"|none)\)\s+\[([^\]]*)\]\s+(\d+)',
                         body):
        nodes.append(dict(order=int(m.group(1)), local_index=int(m.group(2)), node_id=m.group(3), kind="flubble", start=m.group(4),
                          stop=m.group(5), parent=m.group(6), children=strings(m.group(7)), depth=int(m.group(8))))
    for m in re.finditer(r'simpleBoundaryCandidates\s+"([^"]+)"\s+"([^"]+)"', body):
        bounds.append(dict(order=0, node_id="1:1", start=m.group(1), stop=m.group(2)))
    for m in re.finditer(r'simplePvst\s+"([^"]+)"\s+"([^"]+)"', body):
        nodes.append(dict(order=0, node_id="1:0", kind="dummy", start=None, stop=None, parent=None, children=["1:1"], depth=0))
        nodes.append(dict(order=1, local_index=1, node_id="1:1", kind="flubble", start=m.group(1), stop=m.group(2), parent="1:0",
                          children=[], depth=1))
    return dict(gfa=gfa, segments=segs, links=links, boundary_candidates=bounds, pvst_nodes=sorted(nodes, key=lambda n: n["order"]))


_t = lean[lean.index("def fixtureStructureOutput?"):]
table = _t[:_t.index("\ndef ", 5)]
out = {"source": "tests/lean4_conformance/lean_reference.lean fixtureStructureOutput? + src/main.rs cycle_oracle_for_fixture",
       "fixtures": {}}
for m in re.finditer(r'\|\s+"([a-z0-9-]+)"\s+=>\s+some\s+(\([^|]*?\)|\w+)\s*(?=\||$)', table, re.S):
    fid, rhs = m.group(1), m.group(2).strip()
    body = rhs if rhs.startswith("(") else definition(rhs)
    out["fixtures"][fid] = parse_structure(body)

for m in re.finditer(r'"([a-z0-9-]+)"\s+=>\s+Some\(CycleOracle\s*\{\s*fixture_id:\s*"[a-z0-9-]+",\s*equivalent_edge_groups:\s*&\[(.*?)\],\s*\}\)',
                     rs, re.S):
    groups = [[int(x) for x in re.findall(r"\d+", g)] for g in re.findall(r"&\[([^\]]*)\]", m.group(2))]
    out["fixtures"].setdefault(m.group(1), {})["cycle_classes_by_tree_edge_id"] = groups

json.dump(out, open(os.path.join(HERE, "reference_vectors.json"), "w"), indent=1, sort_keys=True)
print("fixtures:", len(out["fixtures"]))
for k, v in sorted(out["fixtures"].items()):
    print(f"  {k}: {v.get('gfa')} segments {len(v.get('segments', []))} links {len(v.get('links', []))} "
          f"boundaries {len(v.get('boundary_candidates', []))} pvst nodes {len(v.get('pvst_nodes', []))} "
          f"classes {v.get('cycle_classes_by_tree_edge_id')}")
