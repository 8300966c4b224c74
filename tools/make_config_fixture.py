#!/usr/bin/env python3
"""(CPU, container only: reads /root/reference) The key / value pairs of the reference's config/default.yaml as a data fixture,
tests/golden/reference_default_config.json -- what tests/test_config_defaults.py holds ygz::Config's built-in defaults and its parser against.
Only values are kept (a flat map key -> string as written), no text of the file."""
import json, os, sys
src = "/root/reference/config/default.yaml"
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_default_config.json")
kv = {}
for ln, line in enumerate(open(src), 1):
    body = line.split("#", 1)[0]
    if body.lstrip().startswith("%") or ":" not in body:
        continue
    k, v = body.split(":", 1)
    k, v = k.strip(), v.strip()
    if k and v:
        kv[k] = {"value": v, "line": ln}
json.dump({"source": "config/default.yaml of PaoPaoRobot/ygz-slam (values only)", "keys": kv}, open(out, "w"), indent=1, sort_keys=True)
print(len(kv), "keys ->", out)
