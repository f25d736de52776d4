import importlib
import os
import sys

import torch
import torch.nn.functional as F


def import_user_module(args):
    """fairseq.utils.import_user_module: the parent of --user-dir goes on sys.path, the directory is imported by its
    basename, then every module of <user-dir>/tasks and <user-dir>/models is imported."""
    module_path = getattr(args, "user_dir", None)
    if module_path is None:
        return
    module_path = os.path.abspath(module_path)
    if not os.path.exists(module_path):
        raise FileNotFoundError(module_path)
    import_user_module.memo = getattr(import_user_module, "memo", set())
    if module_path in import_user_module.memo:
        return
    import_user_module.memo.add(module_path)
    module_parent, module_name = os.path.split(module_path)
    if module_name in sys.modules:
        raise ImportError(f"Failed to import --user-dir={module_path} because the corresponding module name "
                          f"({module_name}) is not globally unique.")
    sys.path.insert(0, module_parent)
    importlib.import_module(module_name)
    for sub in ("tasks", "models"):
        path = os.path.join(module_path, sub)
        if os.path.exists(path):
            for f in sorted(os.listdir(path)):
                if (f.endswith(".py") or os.path.isdir(os.path.join(path, f))) and not f.startswith(("_", ".")):
                    importlib.import_module(f"{module_name}.{sub}." + (f[:-3] if f.endswith(".py") else f))


def softmax(x, dim, onnx_trace=False):
    return F.softmax(x, dim=dim, dtype=torch.float32)


def move_to_cuda(sample, device=None):
    return apply_to_sample(lambda t: t.to(device or "cuda", non_blocking=True), sample)


def apply_to_sample(f, sample):
    """fairseq.utils.apply_to_sample: tensors inside dicts / lists / tuples / sets; anything else is returned as is."""
    def _apply(x):
        if torch.is_tensor(x):
            return f(x)
        if isinstance(x, dict):
            return {k: _apply(v) for k, v in x.items()}
        if isinstance(x, list):
            return [_apply(v) for v in x]
        if isinstance(x, tuple):
            return tuple(_apply(v) for v in x)
        if isinstance(x, set):
            return {_apply(v) for v in x}
        return x
    return _apply(sample)
