"""The micro-benchmarks under tools/ubench/ are evidence sources (DESIGN.md cites their numbers): they have to keep
compiling for gfx950 with the image's hipcc.  Device code only, no GPU needed."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = sorted(glob.glob(os.path.join(ROOT, "tools", "ubench", "*.hip")))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
@pytest.mark.parametrize("src", SOURCES, ids=[os.path.basename(s) for s in SOURCES])
def test_micro_benchmark_compiles_for_gfx950(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    inc = os.path.join(ROOT, "speaker_embedding_ge2e_loss_amd", "csrc")       # mix_split_test checks the kernels' own helper
    out = subprocess.run([hipcc, "-O1", "--offload-arch=gfx950", "-I", inc, "-c", "-o", str(tmp_path / "x.o"), src],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]


def test_every_profile_cited_by_traffic_json_exists():
    """profiles/traffic.json feeds `roofline.traffic` of the bench line and cites the rocprofv3 summary each record came
    from: a citation that points at no committed file is a dangling reference in a driver record."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "traffic.json")) as f:
        table = json.load(f)
    assert table
    for key, rec in table.items():
        src = rec.get("source", "")
        path = src.split(" ", 1)[0]
        assert path.startswith("profiles/") and path.endswith(".txt"), (key, src)
        assert os.path.isfile(os.path.join(root, path)), f"{key}: {path} is cited but not committed"


def test_every_profile_file_cited_in_the_docs_exists():
    """DESIGN.md / README.md / INTEGRATION.md argue with files under profiles/: a citation of a file that is not committed
    is an argument nobody can check."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        with open(os.path.join(root, doc)) as f:
            text = f.read()
        for name in sorted(set(re.findall(r"profiles/([A-Za-z0-9_\-\.]+\.(?:txt|json|csv))", text))):
            if not os.path.isfile(os.path.join(root, "profiles", name)):
                missing.append((doc, name))
    assert not missing, missing
