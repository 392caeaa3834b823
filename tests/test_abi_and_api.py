"""The C-ABI library loads and exports every symbol include/esac_hip.h declares (no compute without a GPU),
and the Python mirror of `esac.forward` keeps the reference's signature and failure convention."""
import ctypes as C
import inspect
import os
import re

import numpy as np
import pytest
import torch

from esac_amd import api, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "esac_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(esac_hip_\w+)\s*\(", src)))


def test_header_and_binding_agree():
    names = _declared_functions()
    assert "esac_hip_forward" in names and "esac_hip_last_error" in names
    assert sorted(api.ABI_SYMBOLS) == names


def test_library_exports_every_declared_symbol():
    path = build.build_hip()
    lib = C.CDLL(path)
    for name in _declared_functions():
        assert hasattr(lib, name), name
    lib.esac_hip_abi_version.restype = C.c_int
    assert lib.esac_hip_abi_version() == api.ABI_VERSION == 6
    # no torch / pybind in the ABI: the shared object must not depend on libtorch or libpython
    import subprocess
    deps = subprocess.run(["ldd", path], capture_output=True, text=True).stdout
    assert "torch" not in deps and "python" not in deps
    assert "amdhip64" in deps
    # RCCL (570 MB) is bound by dlopen at the first esac_hip_comm_* call: a single-GPU process must not load it
    assert "rccl" not in deps


def test_params_struct_layout_matches_header():
    """ctypes mirror of struct esac_hip_params: same field order as the header."""
    src = open(os.path.join(ROOT, "include", "esac_hip.h")).read()
    body = re.search(r"typedef struct esac_hip_params \{(.*?)\} esac_hip_params;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.replace("*", " ").split(None, 1)[1] if not decl.startswith("const") else decl.split("*")[-1]
        fields += [n.strip() for n in names.split(",")]
    assert fields == [f[0] for f in api.Params._fields_]
    assert C.sizeof(api.Params) == 104


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly_not_silently():
    lib = api.load_library()
    ctx = C.c_void_p()
    rc = lib.esac_hip_create(C.byref(ctx), 0)
    assert rc < 0
    assert b"no CPU fallback" in lib.esac_hip_last_error()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        api.Engine(0)
    import esac
    with pytest.raises(RuntimeError):
        esac.forward(torch.zeros(1, 3, 60, 80), torch.zeros(8, dtype=torch.int64), torch.zeros(4, 4),
                     0, 0, 525.0, 320.0, 240.0, 10.0, 100.0, 0.5, 100.0, 8)


def test_forward_signature_matches_reference():
    """esac_forward(sceneCoordinates, hypAssignment, outPose, shiftX, shiftY, focalLength, ppointX, ppointY,
    inlierThreshold, inlierAlpha, inlierBeta, maxReproj, subSampling) -- esac.cpp:64-77."""
    import esac
    params = list(inspect.signature(esac.forward).parameters)
    assert params == ["sceneCoordinates", "hypAssignment", "outPose", "shiftX", "shiftY", "focalLength", "ppointX",
                      "ppointY", "inlierThreshold", "inlierAlpha", "inlierBeta", "maxReproj", "subSampling"]
    assert callable(esac.backward)


def test_backward_signature_matches_reference():
    """esac_backward(sceneCoordinates, outGradients, hypAssignment, gtPose, wLossRot, wLossTrans, lossCut, shiftX,
    shiftY, focalLength, ppointX, ppointY, inlierThreshold, inlierAlpha, inlierBeta, maxReproj, subSampling)
    -- esac.cpp:213-230."""
    import esac
    params = list(inspect.signature(esac.backward).parameters)
    assert params == ["sceneCoordinates", "outGradients", "hypAssignment", "gtPose", "wLossRot", "wLossTrans", "lossCut",
                      "shiftX", "shiftY", "focalLength", "ppointX", "ppointY", "inlierThreshold", "inlierAlpha",
                      "inlierBeta", "maxReproj", "subSampling"]


@pytest.mark.parametrize("bad", ["sc_dtype", "grad_shape", "grad_dtype", "ha_dtype", "gt_shape", "gt_dtype", "empty", "no_gpu"])
def test_backward_argument_validation(bad):
    import esac
    sc = torch.zeros(1, 3, 60, 80)
    g = torch.zeros(1, 3, 60, 80)
    ha = torch.zeros(8, dtype=torch.int64)
    gt = torch.eye(4)
    if bad == "sc_dtype":
        sc = sc.double()
    elif bad == "grad_shape":
        g = torch.zeros(1, 3, 60, 79)
    elif bad == "grad_dtype":
        g = g.double()
    elif bad == "ha_dtype":
        ha = ha.int()
    elif bad == "gt_shape":
        gt = torch.eye(3)
    elif bad == "gt_dtype":
        gt = gt.double()
    elif bad == "empty":
        ha = ha[:0]
    # "no_gpu": valid arguments, but this container has no device -> loud failure, no CPU fallback
    with pytest.raises(RuntimeError):
        esac.backward(sc, g, ha, gt, 1.0, 100.0, 100.0, 0, 0, 525.0, 320.0, 240.0, 10.0, 100.0, 0.5, 100.0, 8)


@pytest.mark.parametrize("bad", ["sc_dtype", "sc_rank", "sc_chan", "ha_dtype", "ha_rank", "pose_shape", "pose_dtype", "empty"])
def test_argument_validation_raises_runtime_error(bad):
    """accessor<float,4>() / accessor<long,1>() / accessor<float,2>() failures surface as RuntimeError (pybind11)."""
    import esac
    sc = torch.zeros(1, 3, 60, 80)
    ha = torch.zeros(8, dtype=torch.int64)
    pose = torch.zeros(4, 4)
    if bad == "sc_dtype":
        sc = sc.double()
    elif bad == "sc_rank":
        sc = sc[0]
    elif bad == "sc_chan":
        sc = torch.zeros(1, 2, 60, 80)
    elif bad == "ha_dtype":
        ha = ha.int()
    elif bad == "ha_rank":
        ha = ha.view(2, 4)
    elif bad == "pose_shape":
        pose = torch.zeros(3, 4)
    elif bad == "pose_dtype":
        pose = pose.double()
    elif bad == "empty":
        ha = ha[:0]
    with pytest.raises(RuntimeError, match="esac.forward"):
        esac.forward(sc, ha, pose, 0, 0, 525.0, 320.0, 240.0, 10.0, 100.0, 0.5, 100.0, 8)


def test_rng_state_helpers():
    import esac
    esac.set_seed(7, 3)
    assert esac.get_rng_state() == (7, 3)
    esac.set_seed(1305, 0)


def test_product_does_not_import_the_oracle():
    """The product path must never route through oracle/ (it is test infrastructure)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "esac_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "esac_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
    for f in ("esac.py",):
        assert "oracle" not in open(os.path.join(ROOT, f)).read()


def test_header_is_plain_c_and_library_resolves_from_c():
    """include/esac_hip.h compiles as C99 with -Wall -Wextra -Werror; a C program dlopens libesac_hip.so, resolves the
    entry points through the header's prototypes and gets the loud no-device failure (no CPU fallback)."""
    import subprocess
    from tests.native import build as nb
    exe = nb.build_abi_check()
    out = subprocess.run([exe, build.LIB_PATH], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "params 104 bytes" in out.stdout and "no CPU fallback" in out.stdout


@pytest.mark.gpu
def test_c_program_runs_forward_and_backward_without_torch():
    """The same C program on the GPU box: hipMalloc'ed buffers through the HIP runtime's C entry points, one forward
    and one backward call -- the boundary really is torch-free."""
    import subprocess
    from tests.native import build as nb
    exe = nb.build_abi_check()
    out = subprocess.run([exe, build.LIB_PATH, "gpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "forward ok" in out.stdout and "backward ok" in out.stdout



@pytest.mark.gpu
def test_c_program_runs_the_several_expert_route_without_torch():
    """Several experts from the plain C caller: the default call takes the speculative route (the library's own streams beside the
    caller's NULL stream), ESAC_DEBUG_NO_SPECULATION the stream order -- the same record and the same score vector, bit for bit."""
    import subprocess
    from tests.native import build as nb
    exe = nb.build_abi_check()
    out = subprocess.run([exe, build.LIB_PATH, "gpu", "spec"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "spec ok" in out.stdout
