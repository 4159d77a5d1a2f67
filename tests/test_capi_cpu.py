"""CPU: the C-ABI library builds/loads, exports every symbol the header declares and
rejects bad arguments before touching the GPU.  No compute is launched here."""
import ctypes
import os
import re

import pytest

from speaker_embedding_ge2e_loss_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.load()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ge2e_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ge2e_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    syms = header_symbols()
    assert "ge2e_loss_fwd_bwd" in syms and "ge2e_cos_sim" in syms and len(syms) >= 8
    raw = ctypes.CDLL(build.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/ge2e_hip.h but not exported"
    assert set(syms) == set(_lib.PROTOTYPES), "ctypes binding and header disagree"


def test_abi_version_and_errors(lib):
    assert lib.ge2e_abi_version() == 2
    assert lib.ge2e_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, -6):
        assert len(lib.ge2e_strerror(code)) > 3


def test_argument_validation_returns_codes_without_gpu(lib):
    f = lib.ge2e_loss_fwd_bwd
    # NULL E
    assert f(None, 1, 4, 5, 8, 16, 16, 1e-8, 1e-6, 0, 0, 16, None, None, None, None, None, 0, None) == -1
    # M = 1 divides by zero in the reference (s3:110-111) -> shape error
    assert f(16, 1, 4, 1, 8, 16, 16, 1e-8, 1e-6, 0, 0, 16, None, None, None, None, 256, 1 << 30, None) == -2
    assert f(16, 0, 4, 5, 8, 16, 16, 1e-8, 1e-6, 0, 0, 16, None, None, None, None, 256, 1 << 30, None) == -2
    # unknown variant / impl
    assert f(16, 1, 4, 5, 8, 16, 16, 1e-8, 1e-6, 7, 0, 16, None, None, None, None, 256, 1 << 30, None) == -4
    assert f(16, 1, 4, 5, 8, 16, 16, 1e-8, 1e-6, 0, 99, 16, None, None, None, None, 256, 1 << 30, None) == -5
    # workspace too small / missing
    assert f(16, 1, 4, 5, 8, 16, 16, 1e-8, 1e-6, 0, 1, 16, None, None, None, None, 256, 8, None) == -3
    assert f(16, 1, 4, 5, 8, 16, 16, 1e-8, 1e-6, 0, 1, 16, None, None, None, None, None, 0, None) == -3
    # dE without dw/db
    assert f(16, 1, 4, 5, 8, 16, 16, 1e-8, 1e-6, 0, 1, 16, None, 32, None, None, 256, 1 << 30, None) == -1
    # misaligned E
    assert f(20, 1, 4, 5, 8, 16, 16, 1e-8, 1e-6, 0, 1, 16, None, None, None, None, 256, 1 << 30, None) == -6
    assert lib.ge2e_cos_sim(None, 1, 4, 5, 8, 1e-8, 1e-6, None, None, 0, None) == -1
    assert lib.ge2e_centroids(None, 1, 4, 5, 8, None, None) == -1
    assert lib.ge2e_calc_loss(None, 1, 4, 5, 1e-6, 0, None, None, None) == -1


def test_workspace_and_impl_queries(lib):
    assert lib.ge2e_workspace_bytes(1, 4, 1, 8, 0, 0) == 0  # bad shape -> 0
    assert lib.ge2e_workspace_bytes(1, 64, 10, 256, 0, 1) > 64 * 256 * 4
    assert lib.ge2e_resolve_impl(1, 64, 10, 256, 0, 1) == 1
    assert lib.ge2e_resolve_impl(1, 64, 10, 256, 0, 0) in (1, 2, 3, 4, 5)
    assert lib.ge2e_workspace_bytes(1, 64, 10, 256, 0, 5) > 8 * 64 * 256 * 4  # team: exchange area, sized without a GPU
    assert lib.ge2e_resolve_impl(1, 64, 1, 256, 0, 0) == -2
    # every impl AUTO can resolve to must report a workspace
    for shape in [(1, 4, 5, 256), (1024, 64, 10, 256), (8, 256, 10, 256), (1, 1024, 10, 768)]:
        impl = lib.ge2e_resolve_impl(*shape, 0, 0)
        assert impl > 0
        assert lib.ge2e_workspace_bytes(*shape, 0, impl) == lib.ge2e_workspace_bytes(*shape, 0, 0)


def test_auto_leaves_the_team_kernel_out_on_request(lib):
    """impl = auto_no_team (a caller that shares the GPU with other streams or processes): AUTO's choice without the
    eight-CU team kernel; an explicit impl=team is still honoured; the workspace query follows the resolved kernel."""
    assert lib.ge2e_resolve_impl(1, 64, 10, 256, 0, _lib.IMPLS["auto"]) == _lib.IMPLS["team"]
    assert lib.ge2e_resolve_impl(1, 64, 10, 256, 0, _lib.IMPLS["auto_no_team"]) == _lib.IMPLS["fused_split"]
    assert lib.ge2e_resolve_impl(1, 64, 10, 256, 0, _lib.IMPLS["team"]) == _lib.IMPLS["team"]
    assert lib.ge2e_resolve_impl(4096, 4, 5, 256, 0, _lib.IMPLS["auto_no_team"]) == _lib.IMPLS["wave"]
    assert lib.ge2e_workspace_bytes(8, 64, 10, 256, 0, _lib.IMPLS["auto_no_team"]) == \
        lib.ge2e_workspace_bytes(8, 64, 10, 256, 0, _lib.IMPLS["fused_split"])


def test_no_environment_override_of_the_implementation_choice():
    """Rounds 2-3 honoured GE2E_AUTO_NO_TEAM=1; the choice is an argument of the call now (impl="auto_no_team") and the
    variable changes nothing."""
    import subprocess
    import sys
    code = ("from speaker_embedding_ge2e_loss_amd import _lib; lib = _lib.load(); "
            "print(lib.ge2e_resolve_impl(1, 64, 10, 256, 0, 0))")
    import os
    env = dict(os.environ, GE2E_AUTO_NO_TEAM="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr
    assert int(out.stdout.strip().splitlines()[-1]) == _lib.IMPLS["team"]


def test_product_refuses_cpu_tensors():
    import torch
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams
    hp = HParams(device="cpu")
    mod = GE2ELoss(hp)
    assert [n for n, _ in mod.named_parameters()] == ["w", "b"]
    assert mod.w.shape == torch.Size([]) and float(mod.w) == 10.0 and float(mod.b) == -5.0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mod(torch.randn(4, 5, 8))


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.GE2ELibraryError, match="not built"):
        _lib.load(str(tmp_path / "nope.so"))


def test_cpp_autograd_library_builds_loads_and_refuses_cpu_tensors():
    """libge2e_torch.so (the autograd node in C++): built next to libge2e_hip.so, registers torch.ops.ge2e_amd.loss, and has
    no CPU path either."""
    import torch
    from speaker_embedding_ge2e_loss_amd import build, functional as GF
    assert os.path.exists(build.build_torch_ext(force=False, verbose=False))
    op = GF._cpp_loss_op()
    assert op is not None
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        op(torch.randn(4, 5, 8), torch.tensor(10.0), torch.tensor(-5.0), 1e-6, 1e-8, 0, 0)


def test_graph_route_module_copies_start_without_captured_graphs():
    """GE2ELoss(hp, graph=True) keeps captured HIP graphs and their static buffers: a copy (deepcopy, pickle) must not drag
    them along (a CUDAGraph cannot be copied) -- it starts empty and captures its own.  Parameters and options copy."""
    import copy
    import pickle
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams
    m = GE2ELoss(HParams(device="cpu"), variant="contrast", graph=True)
    m._steps["k"] = object()          # stands in for a captured step
    m._last_shape = ("k",)
    for c in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert c._steps == {} and c._last_shape is None
        assert c.graph and c.variant == "contrast" and list(c.state_dict()) == ["w", "b"]
    assert "k" in m._steps
