"""
Writes tests/golden/property_rows.json: for every property key of soap_amd/property_table.py, the row of the
reference's table (SOAP/property_table.py `full_property_list`): output name, shape, dtype, unit, dmo_property,
output_physical, a_scale_exponent.  Runs only in the build container (reads /root/reference as text); the
fixture is data, tests/test_host_logic.py compares the package's table with it.
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from soap_amd.property_table import full_property_list  # noqa: E402

text = open("/root/reference/SOAP/property_table.py").read()
entries = dict(re.findall(r'\n        "(\w+)": Property\(\n(.*?)\n        \),', text, re.S))


def field(body, pat):
    m = re.search(pat, body, re.S)
    return m.group(1) if m else None


rows = {}
for key in full_property_list:
    body = entries[key]
    rows[key] = {
        "name": field(body, r'name="(\w+)"'),
        "shape": int(field(body, r"shape=(\d+)")),
        "dtype": field(body, r"dtype=np\.(\w+)"),
        "unit": field(body, r'unit="([^"]*)"').replace(" ", ""),
        "dmo_property": field(body, r"dmo_property=(\w+)") == "True",
        "output_physical": field(body, r"output_physical=(\w+)") == "True",
        "a_scale_exponent": (lambda v: None if v in (None, "None") else int(v))(field(body, r"a_scale_exponent=([\w\-]+)")),
    }
with open(os.path.join(HERE, "property_rows.json"), "w") as f:
    json.dump(rows, f, indent=1, sort_keys=True)
print(len(rows), "rows written")
