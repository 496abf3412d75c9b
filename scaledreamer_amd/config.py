"""Structured configuration without OmegaConf (not installed here).

Mirrors the behaviour the hot path relies on (threestudio/utils/config.py:104-128, utils/misc.py:66-101):
  - parse_structured(Config, cfg): dataclass construction where unknown keys are errors;
  - load_config(yaml, cli_args): YAML + "a.b.c=value" dot-list overrides + ${a.b} interpolation and the
    reference's resolvers, so ScaleDreamer's shipped YAML files load unchanged;
  - C(value, epoch, global_step): scheduled scalars [start_step, start_value, end_value, end_step].
"""
from __future__ import annotations

import dataclasses
import math
import re
from typing import Any, Dict, List, Optional


class ConfigDict(dict):
    """dict with attribute access and .get() — stands in for DictConfig on free-form sub-configs
    (pos_encoding_config, mlp_network_config, loss, optimizer ...)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_config(obj: Any) -> Any:
    if isinstance(obj, dict):
        return ConfigDict({k: to_config(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_config(v) for v in obj]
    return obj


def config_to_primitive(obj: Any, resolve: bool = True) -> Any:
    if isinstance(obj, dict):
        return {k: config_to_primitive(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [config_to_primitive(v) for v in obj]
    if dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        return {f.name: config_to_primitive(getattr(obj, f.name)) for f in dataclasses.fields(obj)}
    return obj


def parse_structured(fields: Any, cfg: Optional[dict] = None) -> Any:
    """Instantiate the dataclass `fields` from `cfg`; an unknown key raises (as OmegaConf.structured does)."""
    cfg = {} if cfg is None else (config_to_primitive(cfg) if not isinstance(cfg, dict) else dict(cfg))
    known = {f.name for f in dataclasses.fields(fields)}
    unknown = sorted(set(cfg) - known)
    if unknown:
        raise KeyError(f"{fields.__qualname__}: unknown config key(s) {unknown}; valid keys: {sorted(known)}")
    return fields(**{k: to_config(v) for k, v in cfg.items()})


# ---- scheduled scalars ---------------------------------------------------------------------------
class Schedule:
    """A scheduled scalar of the configs (threestudio/utils/misc.py:66-101 defines the list format) as a table of knots.
    `[v0, v1, s1]` = from (0, v0) to (s1, v1); `[s0, v0, v1, s1]` = from (s0, v0) to (s1, v1); longer lists append further
    (value, step) pairs: `[s0, v0, v1, s1, v2, s2, ...]`.  The active segment is the last one whose start step has been reached by
    global_step (segment 0 before that); inside it the value is interpolated on global_step when the segment's end step is an int and
    on the epoch otherwise, clamped to the segment."""

    __slots__ = ("steps", "values")

    def __init__(self, spec: Any):
        spec = config_to_primitive(spec)
        if not isinstance(spec, list):
            raise TypeError("Scalar specification only supports list, got", type(spec))
        if len(spec) == 3:
            spec = [0] + spec
        assert len(spec) == 4 or len(spec) >= 6, "a schedule is [v0, v1, s1], [s0, v0, v1, s1] or the latter followed by (value, step) pairs"
        self.steps = [spec[0]] + spec[3::2]
        self.values = ([spec[1]] + spec[2::2])[:len(self.steps)]

    def segment(self, global_step) -> int:
        reached = [k for k in range(1, len(self.steps) - 1) if global_step >= self.steps[k]]
        return reached[-1] if reached else 0

    def at(self, epoch: int, global_step: int, interpolation: str = "linear") -> float:
        k = self.segment(global_step)
        (s0, s1), (v0, v1) = self.steps[k:k + 2], self.values[k:k + 2]
        clock = global_step if isinstance(s1, int) else epoch
        t = min(max((clock - s0) / (s1 - s0), 0.0), 1.0)
        if interpolation == "linear":
            return v0 + (v1 - v0) * t
        if interpolation == "exp":
            return math.exp((1.0 - t) * math.log(v0) + t * math.log(v1))
        raise ValueError(f"Unknown interpolation method: {interpolation}, only support linear and exp")


def C(value: Any, epoch: int, global_step: int, interpolation: str = "linear") -> float:
    """value of a scheduled scalar at (epoch, global_step); plain numbers pass through"""
    if isinstance(value, (int, float)):
        return value
    return Schedule(value).at(epoch, global_step, interpolation)


# ---- YAML loading with ${...} interpolation --------------------------------------------------------
_RESOLVERS = {
    "calc_exp_lr_decay_rate": lambda factor, n: float(factor) ** (1.0 / float(n)),
    "add": lambda a, b: _num(a) + _num(b),
    "sub": lambda a, b: _num(a) - _num(b),
    "mul": lambda a, b: _num(a) * _num(b),
    "div": lambda a, b: _num(a) / _num(b),
    "idiv": lambda a, b: int(_num(a)) // int(_num(b)),
    "basename": lambda p: __import__("os").path.basename(str(p)),
    "rmspace": lambda s, sub: str(s).replace(" ", str(sub)),
    "tuple2": lambda s: [float(s), float(s)],
    "gt0": lambda s: _num(s) > 0,
    "cmaxgt0": lambda s: C_max(s) > 0,
    "not": lambda s: not s,
    "cmaxgt0orcmaxgt0": lambda a, b: C_max(a) > 0 or C_max(b) > 0,
}


def C_max(value: Any) -> float:
    if isinstance(value, (int, float)):
        return value
    value = config_to_primitive(value)
    if len(value) >= 6:
        return max(value[2::2])
    if len(value) == 3:
        value = [0] + value
    return max(value[1], value[2])


def _num(s):
    if isinstance(s, (int, float)):
        return s
    s = str(s)
    try:
        return int(s)
    except ValueError:
        return float(s)


def _lookup(root: dict, dotted: str):
    cur: Any = root
    for part in dotted.split("."):
        cur = cur[int(part)] if isinstance(cur, list) else cur[part]
    return cur


_PAT = re.compile(r"\$\{([^${}]+)\}")


def _resolve_str(root: dict, s: str, depth: int = 0):
    if depth > 20:
        raise ValueError(f"interpolation too deep: {s}")
    m = _PAT.fullmatch(s.strip())

    def one(expr: str):
        expr = expr.strip()
        if ":" in expr:
            name, args = expr.split(":", 1)
            vals = [_resolve_value(root, a.strip(), depth + 1) for a in _split_args(args)]
            return _RESOLVERS[name.strip()](*vals)
        return _resolve_value(root, _lookup(root, expr), depth + 1)

    if m:  # whole string is one interpolation: keep the type
        return one(m.group(1))
    prev = None
    while prev != s and _PAT.search(s):
        prev = s
        s = _PAT.sub(lambda mm: str(one(mm.group(1))), s)
    return s


def _split_args(s: str) -> List[str]:
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
            continue
        depth += ch == "{"
        depth -= ch == "}"
        cur += ch
    out.append(cur)
    return out


def _resolve_value(root: dict, v: Any, depth: int = 0):
    if isinstance(v, str) and "${" in v:
        return _resolve_str(root, v, depth)
    return v


def _resolve_tree(root: dict, node: Any):
    if isinstance(node, dict):
        for k in list(node):
            node[k] = _resolve_tree(root, node[k])
        return node
    if isinstance(node, list):
        return [_resolve_tree(root, v) for v in node]
    return _resolve_value(root, node)


def _set_dotted(root: dict, dotted: str, value: Any):
    cur = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        cur = cur.setdefault(p, {})
    cur[parts[-1]] = value


def load_config(*yamls: str, cli_args: Optional[List[str]] = None, from_string: bool = False, **kwargs) -> ConfigDict:
    """YAML file(s) + dot-list overrides -> resolved ConfigDict (the reference additionally wraps it in
    ExperimentConfig, which only adds trial-directory bookkeeping that this path does not use)."""
    import yaml

    cfg: Dict[str, Any] = {}

    def merge(dst, src):
        for k, v in src.items():
            if isinstance(v, dict) and isinstance(dst.get(k), dict):
                merge(dst[k], v)
            else:
                dst[k] = v

    for y in yamls:
        merge(cfg, yaml.safe_load(y if from_string else open(y)) or {})
    for arg in cli_args or []:
        k, v = arg.split("=", 1)
        _set_dotted(cfg, k, yaml.safe_load(v))
    merge(cfg, kwargs)
    missing = [k for k, v in _walk(cfg) if v == "???"]
    if missing:
        raise ValueError(f"missing mandatory config value(s): {missing}")
    _resolve_tree(cfg, cfg)
    return to_config(cfg)


def _walk(node, prefix=""):
    if isinstance(node, dict):
        for k, v in node.items():
            yield from _walk(v, f"{prefix}{k}.")
    elif isinstance(node, list):
        for i, v in enumerate(node):
            yield from _walk(v, f"{prefix}{i}.")
    else:
        yield prefix[:-1], node
