"""Launcher housekeeping shared by sample.py / sample_refine.py, mirroring the reference's dev_fn/upkeep:

decode_file_macro   dev_fn/upkeep/config.py:26-72 (cb__decode_file with util/subst_util.py:3): list values may contain
                    `?(file:<path>)` entries, which are replaced by the stripped lines of that file (missing file -> nothing);
                    the result is de-duplicated keeping first occurrences.  The reference uses it for --data.process_range.
ckpt_setup          dev_fn/upkeep/ckpt.py:110-123: in commit mode create <cwd>/common/<prog>/<exp_id>/ and mirror the log into
                    its log.txt ("commit mode: setup ckpt"), otherwise "dry run mode"; then log the command line.
ckpt_opt            dev_fn/upkeep/ckpt.py:142-149: dump the resolved options to opt.yml (commit mode, rank 0).
"""
from __future__ import annotations

import logging
import os
import re
import sys
from typing import Dict, Iterable, List, Optional

_logger = logging.getLogger("oakink2_tamf_amd.launch")

_MATCH_SPECIAL = re.compile(r"^\?\((.*)\)$")  # util/subst_util.py:3
_MATCH_FILE = re.compile(r"^file:(.*)$")      # upkeep/config.py:24


def load_fileline(file_name: str) -> List[str]:
    """stripped lines of a text file; a missing file yields no entries (upkeep/config.py:27-38)"""
    path = os.path.normpath(os.path.abspath(file_name))
    if not os.path.exists(path):
        return []
    with open(path, "r") as f:
        return [line.strip() for line in f.read().splitlines()]


def decode_file_macro(values: Optional[Iterable[str]]) -> Optional[List[str]]:
    if values is None:
        return None
    res: List[str] = []
    for el in values:
        m = _MATCH_SPECIAL.fullmatch(el)
        if not m:
            res.append(el)
            continue
        mf = _MATCH_FILE.fullmatch(m.group(1))
        if mf:
            res.extend(load_fileline(mf.group(1)))
        # (an unknown ?(...) command is dropped, as in the reference)
    return list(dict.fromkeys(res).keys())


def ckpt_setup(cfg: Dict, rank: Optional[int] = None, argv: Optional[List[str]] = None) -> None:
    if rank is not None and rank != 0:
        return
    root = logging.getLogger()
    if root.level == logging.NOTSET or root.level > logging.INFO:
        root.setLevel(logging.INFO)  # the launch log is INFO-level (basicConfig is a no-op when the host program configured logging)
    if cfg["commit"]:
        os.makedirs(cfg["ckpt_path"], exist_ok=True)
        cfg["log_file"] = os.path.join(cfg["ckpt_path"], "log.txt")
        handler = logging.FileHandler(cfg["log_file"])
        handler.setFormatter(logging.Formatter("%(asctime)s | %(name)s | %(levelname)s | %(message)s"))
        logging.getLogger().addHandler(handler)
        _logger.info("commit mode: setup ckpt")
    else:
        _logger.info("dry run mode")
    _logger.info("cmd: %s", " ".join([sys.executable] + (sys.argv if argv is None else ["-m", "oakink2_tamf_amd.launch"] + list(argv))))


def ckpt_opt(cfg: Dict, rank: Optional[int] = None) -> None:
    if rank or not cfg["commit"]:
        return
    import yaml

    with open(os.path.join(cfg["ckpt_path"], "opt.yml"), "w") as f:
        yaml.safe_dump({k: v for k, v in cfg.items()}, f, sort_keys=False)
