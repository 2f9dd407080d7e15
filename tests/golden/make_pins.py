#!/usr/bin/env python3
"""Extract the Keras model summaries that the reference's notebooks PRINTED (cell outputs) into a small JSON fixture:
per model the rows (layer name, layer type, output shape, parameter count) and the totals.  These are the only shape-level
known answers the reference holds (it has no tests): tests/test_oracle_pins.py pins the oracle's graph restatement
(oracle.models.keras_summary_*) against them row by row.

Run once in the build container (reads /root/reference; the fixture it writes is data, not source):
    python tests/golden/make_pins.py
"""
import json
import os
import re

REF = "/root/reference/SRModels"
NOTEBOOKS = ["deep_learning_models/ESRGAN.ipynb", "deep_learning_models/SRCNN.ipynb", "deep_learning_models/EDSR.ipynb",
             "defect_detection_models/VGG16.ipynb"]
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "notebook_summaries.json")


def cell_texts(path):
    nb = json.load(open(path))
    for i, c in enumerate(nb["cells"]):
        txt = "".join("".join(o.get("text", [])) for o in c.get("outputs", []) if "text" in o)
        if "Layer (type)" in txt:
            yield i, txt


def parse_model(block):
    """block: text from 'Model: "name"' to the totals.  Fixed-width columns located from the header line; a row whose first column
    continues on the next line (long names / wrapped shapes) is joined."""
    lines = block.splitlines()
    hdr = next(l for l in lines if "Layer (type)" in l)
    c_shape, c_par = hdr.index("Output Shape"), hdr.index("Param #")
    c_conn = hdr.index("Connected to") if "Connected to" in hdr else len(hdr) + 200
    rows, cur = [], None
    started = False
    for l in lines[lines.index(hdr) + 1:]:
        if set(l.strip()) <= set("=_") and l.strip():
            if started and l.strip().startswith("=") and rows:
                break                                         # the rule above the totals
            started = True
            continue
        if not l.strip():
            continue
        a, b, c = l[:c_shape].strip(), l[c_shape:c_par].strip(), l[c_par:c_conn].strip()
        if c.isdigit() and a:                                 # first line of a row
            cur = [a, b, int(c)]
            rows.append(cur)
        elif cur is not None:                                 # continuation
            cur[0] += a
            cur[1] += (" " if b and not cur[1].endswith(",") else "") + b if b else ""
    out = []
    for a, b, n in rows:
        m = re.match(r"^(\S+)\s*\((.*)\)$", a.replace(" ", ""))
        name, typ = (m.group(1), m.group(2)) if m else (a, "")
        dims = [None if t.strip() == "None" else int(t) for t in re.sub(r"[\[\]()]", "", b).split(",") if t.strip()]
        out.append([name, typ, dims, n])
    tot = {k: int(v.replace(",", "")) for k, v in re.findall(r"(Total|Trainable|Non-trainable) params: ([\d,]+)", block)}
    return {"rows": out, "totals": tot}


def main():
    res = {}
    for nbp in NOTEBOOKS:
        for cell, txt in cell_texts(os.path.join(REF, nbp)):
            for block in txt.split('Model: "')[1:]:
                name = block.split('"')[0]
                res[name] = dict(parse_model(block), source=f"{nbp} cell {cell}")
    json.dump(res, open(OUT, "w"), indent=0, separators=(",", ":"))
    for k, v in res.items():
        print(k, len(v["rows"]), v["totals"], v["source"])


if __name__ == "__main__":
    main()
