"""CPU test: ygz::Config of the class surface (include/ygz/Basic/Config.h, ygz_slam_amd/host/ygz_host.cpp) against the VALUES of the reference's
config/default.yaml, held as a data fixture (tests/golden/reference_default_config.json, made by tools/make_config_fixture.py from the reference
in the build container).  Two things are pinned to reference-held data: (a) the defaults built into the surface -- every numeric key of the file
reads the file's value without any parameter file; (b) the "key: value" reader -- the same pairs written back as a parameter file, with comments
and the %YAML header as cv::FileStorage files carry them, read to the same values.  The driver is tests/cpp/test_surface in its `config` mode,
which touches no device."""
import json
import os
import subprocess
import pytest
from conftest import ROOT

BIN = os.path.join(ROOT, "tests", "cpp", "test_surface")
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_default_config.json")))["keys"]


def _run(args):
    if not os.path.exists(BIN):
        pytest.skip("tests/cpp/test_surface not built (python -c 'import __graft_entry__ as g; g.build()')")
    out = subprocess.run([BIN, "config"] + args, capture_output=True, text=True, check=True).stdout
    got = {}
    for line in out.splitlines():
        k, raw, num = line.split(" ")
        got[k] = (raw, float(num))
    return got


def _numeric(v):
    try:
        return float(v)
    except ValueError:
        return None


def test_built_in_defaults_are_the_reference_files_values():
    num = {k: _numeric(e["value"]) for k, e in FIX.items() if _numeric(e["value"]) is not None}
    assert len(num) >= 23 and {"matcher.th_low", "feature.cell", "camera.fx", "frame.pyramid", "LocalMapping.local_mappoints"} <= set(num)
    got = _run(["--"] + sorted(num))
    for k, v in num.items():
        assert got[k][1] == v, (k, got[k], v)
    # a key the file does not hold reads 0 (the reference's cv::FileStorage returns an empty node, which converts to 0)
    assert _run(["--", "camera.k1"])["camera.k1"][1] == 0.0


def test_parameter_file_reader_on_the_reference_files_pairs(tmp_path):
    f = tmp_path / "params.yaml"
    lines = ["%YAML:1.0", "# comment line", ""]
    changed = {}
    for i, (k, e) in enumerate(sorted(FIX.items())):
        v = e["value"]
        if _numeric(v) is not None and i % 3 == 0:            # some values changed: the file must win over the built-in default
            v = repr(_numeric(v) * 2 + 1)
            changed[k] = float(v)
        lines.append("%s: %s   # trailing comment" % (k, v) if i % 2 else "%s:%s" % (k, v))
    f.write_text("\n".join(lines) + "\n")
    keys = sorted(FIX)
    got = _run([str(f), "--"] + keys)
    for k in keys:
        want = changed.get(k, _numeric(FIX[k]["value"]))
        if want is None:
            assert got[k][0] == FIX[k]["value"].split()[0] and got[k][1] == 0.0      # a string value: kept as written, Get<double> reads 0
        else:
            assert got[k][1] == want, (k, got[k], want)
    r = subprocess.run([BIN, "config", str(tmp_path / "missing.yaml"), "--", "image.width"], capture_output=True, text=True)
    assert r.returncode == 3                                   # Config::SetParameterFile returns false (Config.cpp:6-18 logs and keeps going)
