"""The drop-in boundary, literally: with `dropin/` ahead on sys.path, every import statement the four reference
trainers make of a module of the reference tree binds this package, every name the trainers then use resolves, and every
function / class among them keeps the reference's parameter list.

The surface is DERIVED from the reference's source, not hand-picked: oracle/gen_golden.py g9 walks the ASTs of
train_arco_2d.py / train_arco_3d.py / pretrain_2D.py / pretrain_3D.py, collects each `from <reference module> import ...`
statement, the free names only a star import can supply (module aliases like np / F / nn included), the attributes read from
imported sub-modules (`losses.DiceLoss`, `ramps.sigmoid_rampup`), and the AST signatures of the functions / classes
(tests/golden/g9_flags.json: "trainer_surface", "surface_signatures").  CPU-only; each trainer runs in a fresh interpreter
so the top-level module names (`utils`, `augment`, `model_2D`, ...) never leak into this process."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G9 = json.load(open(os.path.join(ROOT, "tests", "golden", "g9_flags.json")))

_CHILD = r'''
import inspect, json, sys
sys.path.insert(0, sys.argv[1])
spec = json.loads(sys.stdin.read())
ns, report = {}, {"missing": [], "sig": [], "files": []}
for e in spec["imports"]:
    exec(e["stmt"], ns)                                     # the reference's own import statement, verbatim
    top = e["module"].split(".")[0]
    for name, kind in e["names"].items():
        if isinstance(kind, dict):                          # `from pkg import mod`: the attributes the trainer reads
            mod = ns[name]
            report["files"].append(getattr(mod, "__file__", "?"))
            for a in kind["module_attrs"]:
                if not hasattr(mod, a):
                    report["missing"].append(f"{e['module']}.{name}.{a}")
                else:
                    ns[f"{e['module']}.{name}.{a}"] = getattr(mod, a)
        elif name not in ns:
            report["missing"].append(f"{e['module']}:{name}")
        else:
            ns[f"{e['module']}.{name}"] = ns[name]
    m = sys.modules.get(e["module"])
    if m is not None and getattr(m, "__file__", None):
        report["files"].append(m.__file__)
for key, ref in spec["signatures"].items():
    o = ns.get(key)
    if o is None:
        continue
    ps = list(inspect.signature(o.__init__ if inspect.isclass(o) else o).parameters.values())
    mine = [[q.name, None if q.default is inspect.Parameter.empty else repr(q.default)] for q in ps]
    if mine[:len(ref)] != ref or any(d is None for _, d in mine[len(ref):]):
        report["sig"].append([key, mine, ref])
report["arco"] = sorted(k for k, v in sys.modules.items() if k.startswith("arco_amd"))[:3]
print(json.dumps(report))
'''


@pytest.mark.parametrize("trainer", sorted(G9["trainer_surface"]))
def test_reference_import_statements_bind_this_package(trainer):
    surf = G9["trainer_surface"][trainer]
    assert surf["unresolved"] == []                       # the reference itself resolves every free name through its stars
    assert len(surf["imports"]) >= 5
    spec = dict(imports=surf["imports"], signatures=G9["surface_signatures"])
    env = dict(os.environ, PYTHONPATH="")
    r = subprocess.run([sys.executable, "-c", _CHILD, os.path.join(ROOT, "dropin")], input=json.dumps(spec), text=True,
                       capture_output=True, cwd="/", env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["missing"] == [], rep["missing"]
    assert rep["sig"] == [], rep["sig"]
    assert rep["arco"], "the import statements did not load arco_amd"
    # every reference-tree module the trainer imported came from dropin/ (-> arco_amd), none from anywhere else
    for f in rep["files"]:
        assert os.path.dirname(os.path.abspath(f)).startswith(os.path.join(ROOT, "dropin")), f


def test_surface_covers_the_names_the_verdict_named():
    """LocalConLoss (train_arco_2d.py:270), utils.losses.DiceLoss (:17,269), ramps.sigmoid_rampup (:123) are in the derived list."""
    t2 = {e["module"]: e["names"] for e in G9["trainer_surface"]["train_arco_2d.py"]["imports"]}
    assert "LocalConLoss" in t2["loss_helper_3d"] and "compute_contra_memobank_loss" in t2["loss_helper_3d"]
    assert t2["utils"]["losses"]["module_attrs"] == ["DiceLoss"] and t2["utils"]["ramps"]["module_attrs"] == ["sigmoid_rampup"]
    assert {"batch_transform", "generate_unsup_data", "randomGeneratorWithLogits"} <= set(t2["augment"])
    assert {"np", "nn", "F"} <= set(t2["loss_helper_3d"])                 # aliases the trainer never imports itself


@pytest.mark.skipif(not os.path.isdir("/root/reference/code"), reason="the reference tree exists in the build container only")
@pytest.mark.parametrize("trainer", ["train_arco_2d.py", "train_arco_3d.py", "pretrain_2D.py", "pretrain_3D.py"])
def test_reference_trainer_import_block_runs_against_dropin(trainer):
    """Build container only: the REAL import block of the reference trainer (every top-level import statement of the file,
    in order, third-party packages absent from this image stubbed) executed with PYTHONPATH = dropin : code, then every
    name the trainer's code loads is looked up in the resulting namespace - what `python train_arco_2d.py` does up to
    its first statement."""
    child = r'''
import ast, builtins, sys, types
trainer, dropin, code = sys.argv[1:4]
sys.path[:0] = [dropin, code]
class _Stub(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Stub(self.__name__ + "." + n)
    def __call__(self, *a, **k):
        return None
for name in ("tensorboardX", "torchvision", "torchvision.transforms", "torchvision.utils", "torchvision.models", "h5py", "medpy",
             "SimpleITK", "nibabel", "skimage", "skimage.measure"):
    try:
        __import__(name)
    except Exception:
        sys.modules[name] = _Stub(name)
src = open(code + "/" + trainer).read()
tree = ast.parse(src)
ns = {"__name__": "trainer_imports"}
for node in tree.body:
    if isinstance(node, (ast.Import, ast.ImportFrom)):
        exec(compile(ast.Module([node], []), trainer, "exec"), ns)
bound = set()
for n in ast.walk(tree):
    if isinstance(n, ast.Name) and not isinstance(n.ctx, ast.Load): bound.add(n.id)
    elif isinstance(n, (ast.FunctionDef, ast.ClassDef)): bound.add(n.name)
    elif isinstance(n, ast.arg): bound.add(n.arg)
loads = {n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
missing = sorted(n for n in loads if n not in ns and n not in bound and not hasattr(builtins, n))
from_ref = sorted(k for k, m in sys.modules.items() if getattr(m, "__file__", None) and m.__file__.startswith(code)
                  and not k.startswith(("networks.", "utils.util", "dataloaders.utils", "tps_stn")))
print("MISSING", missing)
print("FROM_REF", from_ref)
'''
    r = subprocess.run([sys.executable, "-c", child, trainer, os.path.join(ROOT, "dropin"), "/root/reference/code"],
                       capture_output=True, text=True, cwd="/", env=dict(os.environ, PYTHONPATH=""), timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = dict(l.split(" ", 1) for l in r.stdout.strip().splitlines() if l.startswith(("MISSING", "FROM_REF")))
    assert out["MISSING"] == "[]", out
    # hot-path modules must come from dropin/, not from the reference tree (dataloaders.utils, which the stage-1 trainers
    # import and never use, is allowed to fall through to the reference's own file: namespace packages merge)
    assert out["FROM_REF"] == "[]", out


_SHADOW_CHILD = r'''
import sys
sys.path.insert(0, sys.argv[1])
# what the reference trainers bind BEFORE their star imports (train_arco_2d.py:1-19, train_arco_3d.py:1-19)
import torch, logging, os, pickle, argparse, shutil, random
import torch.optim as optim
from torch.nn.modules.loss import CrossEntropyLoss
import torch.utils.data.sampler as sampler
from utils import losses, metrics, ramps
before = dict(torch=torch, logging=logging, sys=sys, os=os, pickle=pickle, optim=optim, argparse=argparse, shutil=shutil,
              random=random, CrossEntropyLoss=CrossEntropyLoss, sampler=sampler, losses=losses, metrics=metrics, ramps=ramps)
ns = dict(before)
for stmt in sys.argv[2:]:
    exec(stmt, ns)
bad = [k for k, v in before.items() if ns.get(k) is not v]
print("SHADOWED " + ",".join(bad))
'''


@pytest.mark.parametrize("stmts", [["from augment import *", "from loss_helper_3d import *", "from model_2D import *"],
                                   ["from augment_3d import *", "from loss_helper import *", "from model_3D import *"]])
def test_star_imports_do_not_rebind_what_the_trainer_imported_before(stmts):
    """Round 4, found by tests/test_dropin_loop_gpu.py: `arco_amd.model_2D` bound its optimiser module as `optim`, and the
    reference trainers do `import torch.optim as optim` (train_arco_2d.py:8) BEFORE `from model_2D import *` (:24) - the star
    import replaced torch.optim and `optim.SGD(...)` (:248) raised.  No name the trainers bind ahead of their star imports may be
    rebound by them."""
    out = subprocess.run([sys.executable, "-c", _SHADOW_CHILD, os.path.join(ROOT, "dropin")] + stmts, capture_output=True, text=True,
                         timeout=300, env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"})
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("SHADOWED")][-1]
    assert line.strip() == "SHADOWED", line


def test_star_imported_nn_is_torch_nn_with_two_subclasses():
    """The `nn` the reference trainers receive through `from model_2D import *` / `from loss_helper_3d import *` (they never import
    torch.nn; train_arco_2d.py:158,231-234,419): torch.nn itself, except Conv2d / Conv3d subclasses whose GPU 1x1 forward takes the HIP path
    (arco_amd/nn_dropin.py).  On a CPU-only host every call falls through to torch's own forward, bit for bit."""
    code = r'''
import sys, json
sys.path.insert(0, sys.argv[1])
import torch
from model_2D import *
from loss_helper_3d import *
tnn = torch.nn
q = nn.Sequential(nn.Conv2d(32, 32, kernel_size=1, bias=False), nn.Conv2d(32, 32, kernel_size=1, bias=False))
x = torch.randn(2, 32, 5, 7)
y = q(x)
ref = tnn.functional.conv2d(tnn.functional.conv2d(x, q[0].weight), q[1].weight)
v = nn.Conv3d(16, 16, kernel_size=1, bias=False)
print("NN " + json.dumps(dict(name=[type(q[0]).__name__, type(v).__name__], sub=[isinstance(q[0], tnn.Conv2d), isinstance(v, tnn.Conv3d)],
      same=bool(torch.equal(y, ref)), keys=sorted(q.state_dict()), rest=[nn.Sequential is tnn.Sequential, nn.KLDivLoss is tnn.KLDivLoss,
      nn.functional.normalize is tnn.functional.normalize, nn.Module is tnn.Module, nn.BatchNorm2d is tnn.BatchNorm2d],
      module=type(q[0]).__module__, repr=repr(q[0]))))
'''
    for flag, mod in (("1", "arco_amd.nn_dropin"), ("0", "torch.nn.modules.conv")):
        env = dict({k: v for k, v in os.environ.items() if k != "PYTHONPATH"}, ARCO_DROPIN_NN=flag)
        r = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "dropin")], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("NN ")][-1][3:])
        assert out["name"] == ["Conv2d", "Conv3d"] and out["sub"] == [True, True] and out["same"] and out["keys"] == ["0.weight", "1.weight"]
        assert all(out["rest"]) and out["module"] == mod
        assert out["repr"] == "Conv2d(32, 32, kernel_size=(1, 1), stride=(1, 1), bias=False)"
