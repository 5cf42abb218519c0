"""YAML + command-line configuration surface of the pipeline -- same grammar and keys as the reference's
`options.py` (parse_arguments :23-46, set :48-60, load_options :62-76, override_options :78-95,
process_options :97-113, save_options_file :116-137):

    --yaml=configs/reconstruct/<case>      (mandatory, path without .yaml)
    --a.b.c=value | --flag | --flag! | --key=     (True / False / None)

`_parent_` inheritance is kept.  Differences, all about never blocking a batch job: the two interactive
`input()` prompts of the reference (unknown key, differing options.yaml) are answered "y" automatically unless
stdin is a TTY.  opt.device is "cuda:<gpu>" when a HIP device is visible; `--cpu` is accepted for
compatibility but the MI355X path refuses to run on it (there is no CPU fallback in this package).
"""
import os
import random
import string
import sys

import numpy as np
import yaml


class Opt(dict):
    """dict with attribute access (stands in for easydict.EasyDict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]


def to_opt(d):
    if isinstance(d, dict):
        return Opt({k: to_opt(v) for k, v in d.items()})
    return d


def to_dict(d):
    if isinstance(d, dict):
        return {k: to_dict(v) for k, v in d.items()}
    if isinstance(d, np.ndarray):
        return d.tolist()
    return d


def parse_arguments(args):
    opt_cmd = {}
    for arg in args:
        assert arg.startswith("--"), arg
        body = arg[2:]
        if "=" not in body:
            key_str, value = (body[:-1], "false") if body.endswith("!") else (body, "true")
        else:
            key_str, value = body.split("=", 1)
        keys = key_str.split(".")
        sub = opt_cmd
        for k in keys[:-1]:
            sub = sub.setdefault(k, {})
        assert keys[-1] not in sub, keys[-1]
        sub[keys[-1]] = yaml.safe_load(value)
    return to_opt(opt_cmd)


def _ask(question):
    if sys.stdin is not None and sys.stdin.isatty():
        ans = None
        while ans not in ("y", "n"):
            ans = input(question)
        return ans
    print(question + "y   [non-interactive: auto-accepted]")
    return "y"


def override_options(opt, opt_over, key_stack=None, safe_check=False):
    key_stack = key_stack or []
    for key, value in opt_over.items():
        if isinstance(value, dict):
            opt[key] = override_options(opt.get(key, Opt()), value, key_stack + [key], safe_check)
        else:
            if safe_check and key not in opt:
                if _ask('"%s" not found in original opt, add? (y/n) ' % ".".join(key_stack + [key])) == "n":
                    print("safe exiting...")
                    sys.exit()
            opt[key] = value
    return opt


def load_options(fname):
    with open(fname) as f:
        opt = to_opt(yaml.safe_load(f))
    if "_parent_" in opt:
        parents = opt.pop("_parent_")
        parents = [parents] if isinstance(parents, str) else parents
        for p in parents:
            opt = override_options(load_options(p), opt, key_stack=[])
    print("loading {}...".format(fname))
    return opt


def process_options(opt):
    import torch

    if opt.get("seed") is not None:
        random.seed(opt.seed)
        np.random.seed(opt.seed)
        torch.manual_seed(opt.seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(opt.seed)
        if opt.seed != 0:
            opt.name = str(opt.name) + "_seed{}".format(opt.seed)
    else:
        # unseeded run: the reference appends four random letters to the run name (options.py:106-107).  Under
        # torch.distributed every rank must end up with the SAME name (ranks > 0 read files rank 0 writes) and the
        # same numpy stream (load_colmap_points jitters the candidates with np.random): rank 0 draws, all adopt.
        suffix = "".join(random.choice(string.ascii_uppercase) for _ in range(4))
        shared_seed = random.randrange(2 ** 31)
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            import torch.distributed as tdist

            if tdist.is_available() and tdist.is_initialized():
                box = [suffix, shared_seed]
                tdist.broadcast_object_list(box, src=0)
                suffix, shared_seed = box
            else:
                # called before init_process_group (tools, tests, infer_inner.get_config): every rank derives the SAME
                # values from what the launcher gave all of them instead of failing.  The rendezvous id / address / port are
                # the same for every launch with torchrun's static defaults ("none", 127.0.0.1, 29500), so under torchrun
                # (TORCHELASTIC_RUN_ID set: all ranks of a node are children of ONE agent process) the agent's pid and the
                # restart count are mixed in -- distinct per launch, identical across the ranks.  Under any other launcher
                # nothing per-launch is known to be shared: the values are then a pure function of the environment, and the
                # run says so (the reference draws a fresh suffix per run; give --name / --seed to choose).
                import hashlib
                import warnings

                key = "%s|%s|%s" % (os.environ.get("TORCHELASTIC_RUN_ID", ""), os.environ.get("MASTER_ADDR", ""),
                                    os.environ.get("MASTER_PORT", ""))
                if "TORCHELASTIC_RUN_ID" in os.environ:
                    key += "|%s" % os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
                    # the agent's pid makes the values differ from launch to launch -- but only ranks that are direct
                    # children of ONE agent share it: one node (every rank local), no wrapper process in between
                    one_agent = os.environ.get("LOCAL_WORLD_SIZE") == os.environ.get("WORLD_SIZE") and \
                        os.environ.get("GROUP_WORLD_SIZE", "1") == "1"
                    if one_agent and not os.environ.get("MH_NO_PPID_SEED"):
                        key += "|%d" % os.getppid()
                    else:
                        warnings.warn("options: unseeded multi-node (or wrapped) torchrun launch without an initialised "
                                      "process group: the output-name suffix and the candidate jitter are derived from the "
                                      "rendezvous id / address / port only (give --name / --seed, or initialise the process "
                                      "group first)")
                else:
                    warnings.warn("options: unseeded multi-rank run without an initialised process group and without "
                                  "torchrun: the output-name suffix and the candidate jitter are derived from "
                                  "MASTER_ADDR/MASTER_PORT only and repeat from launch to launch")
                h = hashlib.sha256(key.encode()).digest()
                suffix = "".join(string.ascii_uppercase[b % 26] for b in h[:4])
                shared_seed = int.from_bytes(h[4:8], "little") % (2 ** 31)
            # (the numpy stream is re-seeded only here: a single-rank unseeded run keeps numpy's own entropy, as the
            # reference does; multi-rank runs need the ranks to draw the same candidate jitter)
            np.random.seed(shared_seed)
        opt.name = str(opt.name) + "_" + suffix
    assert isinstance(opt.gpu, int)
    local = os.environ.get("MH_DEVICE_OVERRIDE", os.environ.get("LOCAL_RANK"))
    gpu = int(local) if local is not None else opt.gpu      # one process per GPU under torchrun
    opt.device = "cpu" if opt.get("cpu") or not torch.cuda.is_available() else "cuda:{}".format(gpu)


def set(opt_cmd={}):
    assert "yaml" in opt_cmd, "--yaml=<config path without .yaml> is mandatory"
    opt = load_options("{}.yaml".format(opt_cmd.yaml))
    opt = override_options(opt, opt_cmd, key_stack=[], safe_check=True)
    process_options(opt)
    return opt


def save_options_file(opt):
    fname = "{}/options.yaml".format(opt.output_path)
    cur = to_dict(opt)
    if os.path.isfile(fname):
        with open(fname) as f:
            old = yaml.safe_load(f)
        if cur != old:
            print("existing options file found (different from current one)...")
            if _ask("override? (y/n) ") == "n":
                print("safe exiting...")
                sys.exit()
        else:
            print("existing options file found (identical)")
    else:
        print("(creating new options file...)")
    with open(fname, "w") as f:
        yaml.safe_dump(cur, f, default_flow_style=False, indent=4)
