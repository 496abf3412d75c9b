"""Plugin registry with the semantics of the reference's threestudio.register / threestudio.find
(threestudio/__init__.py:5-32): unique names, and "main:mixin1,mixin2" composition in find()."""
from __future__ import annotations

import logging
from typing import Dict

__modules__: Dict[str, type] = {}
logger = logging.getLogger("scaledreamer_amd")


def register(name: str):
    def decorator(cls):
        if name in __modules__:
            raise ValueError(f"Module {name} already exists! Names of extensions conflict!")
        __modules__[name] = cls
        return cls

    return decorator


def find(name: str) -> type:
    if ":" in name:
        main_name, sub_name = name.split(":")
        parts = sub_name.split(",") if "," in sub_name else [sub_name]
        parts.append(main_name)
        return type(f"{main_name}.{sub_name}", tuple(__modules__[n] for n in parts), {})
    return __modules__[name]


def _rank() -> int:
    import os

    return int(os.environ.get("RANK", os.environ.get("LOCAL_RANK", "0")))


def info(*a, **k):
    if _rank() == 0:
        logger.info(*a, **k)


def debug(*a, **k):
    if _rank() == 0:
        logger.debug(*a, **k)


def warn(*a, **k):
    if _rank() == 0:
        logger.warning(*a, **k)
