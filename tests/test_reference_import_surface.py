"""Every `from verl... import name` the REFERENCE's own modules make (its internal import graph = the names third-party code written
against it can reach) resolves in this build, except an explicit list of names that belong to subsystems this build replaces.
Reads /root/reference with `ast` only (nothing is imported from it); skipped where the reference tree is absent (the GPU box)."""
import ast
import importlib
import os

import pytest

REF = "/root/reference"

# module -> names that are deliberately absent, with the reason
OUT_OF_SCOPE = {
    # Ray control plane: this build runs one torchrun process per GPU (verl/single_controller/base/worker_group.py drives them in-process)
    "verl.single_controller.ray": "*", "verl.single_controller.ray.base": "*", "verl.single_controller.base.register_center.ray": "*",
    "verl.trainer.ray_trainer": {"ResourcePoolManager"}, "verl.protocol": {"DataProtoFuture"},
    "verl.single_controller.base.decorator": {"get_predefined_dispatch_fn", "get_predefined_execute_fn"},
    # vLLM rollout engine and its weight-resharding manager: generation is spatialthinker_amd.rollout.Generator on the actor's own weights
    "verl.workers.rollout": {"vLLMRollout"}, "verl.workers.rollout.vllm_rollout_spmd": "*",
    "verl.workers.sharding_manager": {"FSDPVLLMShardingManager"}, "verl.workers.sharding_manager.fsdp_vllm": "*",
    # torch FSDP wrapping / offload helpers and the monkey patches of HF attention modules: there is no HF nn.Module in this build
    "verl.utils.fsdp_utils": "*", "verl.models.monkey_patch": "*", "verl.models.transformers.flash_attention_utils": "*",
    "verl.models.transformers.qwen2_vl": {"qwen2_vl_attn_forward"},
}


def _reference_import_graph():
    want = {}
    for dp, _dn, files in os.walk(os.path.join(REF, "verl")):
        for f in files:
            if not f.endswith(".py"):
                continue
            path = os.path.join(dp, f)
            parts = os.path.relpath(path, REF)[:-3].split(os.sep)
            pkg = parts[:-1]                                  # the package a relative import is resolved against
            for node in ast.walk(ast.parse(open(path).read())):
                if not isinstance(node, ast.ImportFrom):
                    continue
                if node.level:
                    base = pkg[:len(pkg) - node.level + 1]
                    mod = ".".join(base + ([node.module] if node.module else []))
                elif node.module and node.module.split(".")[0] == "verl":
                    mod = node.module
                else:
                    continue
                want.setdefault(mod, set()).update(a.name for a in node.names if a.name != "*")
    return want


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "verl")), reason="the reference tree only exists in the build container")
def test_the_references_internal_imports_resolve_here():
    want = _reference_import_graph()
    assert len(want) >= 50
    missing, stale = [], []
    for mod, names in sorted(want.items()):
        skip = OUT_OF_SCOPE.get(mod, set())
        if skip == "*":
            continue
        try:
            m = importlib.import_module(mod)
        except ImportError as e:
            missing.append(f"{mod}: {e}")
            continue
        for n in sorted(names):
            ok = hasattr(m, n)
            if not ok:
                try:
                    importlib.import_module(f"{mod}.{n}")
                    ok = True
                except ImportError:
                    pass
            if n in skip:
                if ok:
                    stale.append(f"{mod}.{n} is listed out of scope but exists")
            elif not ok:
                missing.append(f"{mod}.{n}")
    assert not missing, missing
    assert not stale, stale


# public functions / methods of the reference that are deliberately absent or differ, with the reason
SIGNATURE_OUT_OF_SCOPE = {
    # Ray control plane (futures, resource pools, dispatch / collect tables of the Ray worker group, aliveness watchdog)
    "verl.protocol": {"DataProtoFuture.concat", "DataProtoFuture.chunk", "DataProtoFuture.get"},
    "verl.trainer.main": {"Runner.run"},
    "verl.trainer.ray_trainer": {"ResourcePoolManager.create_resource_pool", "ResourcePoolManager.get_resource_pool", "ResourcePoolManager.get_n_gpus",
                                 "RayPPOTrainer.__init__"},          # takes this build's in-process worker groups instead of role -> Ray class maps
    "verl.single_controller.base.worker": {"WorkerHelper.get_availale_master_addr_port", "WorkerMeta.to_dict", "WorkerMeta.__init__"},
    "verl.single_controller.base.decorator": {"dispatch_one_to_all", "dispatch_all_to_all", "collect_all_to_all", "dispatch_dp_compute", "collect_dp_compute",
                                              "dispatch_dp_compute_data_proto", "dispatch_dp_compute_data_proto_with_func", "collect_dp_compute_data_proto",
                                              "get_predefined_dispatch_fn", "get_predefined_execute_fn"},
    "verl.single_controller.base.worker_group": {"check_workers_alive", "WorkerGroup.start_worker_aliveness_check", "WorkerGroup.world_size"},   # world_size is an attribute here
    # the HF attention monkey patch
    "verl.models.transformers.qwen2_vl": {"qwen2_vl_attn_forward"},
    # internals of the Karmarkar-Karp implementation and of the YAML float formatting
    "verl.utils.seqlen_balancing": {"Set.add", "Set.merge", "Set.__init__", "State.get_partitions", "State.merge", "State.spread", "State.__init__"},
    "verl.utils.py_functional": {"is_sci_notation", "float_representer"},
}


def _reference_public_defs(path):
    out = {}
    tree = ast.parse(open(path).read())
    params = lambda fn: [a.arg for a in fn.args.posonlyargs + fn.args.args + fn.args.kwonlyargs if a.arg not in ("self", "cls")]
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and not node.name.startswith("_"):
            out[node.name] = params(node)
        elif isinstance(node, ast.ClassDef) and not node.name.startswith("_"):
            for b in node.body:
                if isinstance(b, ast.FunctionDef) and (not b.name.startswith("_") or b.name in ("__init__", "__call__")):
                    out[f"{node.name}.{b.name}"] = params(b)
    return out


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "verl")), reason="the reference tree only exists in the build container")
def test_public_functions_keep_the_references_parameter_names():
    """For every module that exists on both sides: each public function / method of the reference exists here and accepts the
    reference's parameter names (extra parameters with defaults are fine)."""
    import inspect
    problems, stale, checked = [], [], 0
    for dp, _dn, files in os.walk(os.path.join(REF, "verl")):
        for f in files:
            if not f.endswith(".py"):
                continue
            path = os.path.join(dp, f)
            mod = os.path.relpath(path, REF)[:-3].replace(os.sep, ".")
            mod = mod[:-9] if mod.endswith(".__init__") else mod
            try:
                m = importlib.import_module(mod)
            except ImportError:
                continue                                     # whole modules out of scope: listed in OUT_OF_SCOPE above
            skip = SIGNATURE_OUT_OF_SCOPE.get(mod, set())
            for name, want in _reference_public_defs(path).items():
                obj, found = m, True
                for part in name.split("."):
                    if not hasattr(obj, part):
                        found = False
                        break
                    obj = getattr(obj, part)
                ok = found
                if found:
                    try:
                        have = [p for p in inspect.signature(obj).parameters if p not in ("self", "cls")]
                        ok = set(want) <= set(have) or have[:len(want)] == want
                    except (TypeError, ValueError):
                        ok = True
                checked += 1
                if name in skip:
                    if ok and found and name != "WorkerGroup.world_size":
                        stale.append(f"{mod}:{name} is listed out of scope but matches")
                elif not ok:
                    problems.append(f"{mod}:{name} wants {want}" + ("" if found else " (missing)"))
    assert checked > 250
    assert not problems, problems
    assert not stale, stale
