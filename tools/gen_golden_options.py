"""Golden vectors for the configuration surface (options.py of the reference: parse_arguments :23-46, set :48-60,
load_options :62-76 with `_parent_` chains, override_options :78-95, process_options :97-113):
    python tools/gen_golden_options.py        (build container only; writes tests/golden/options.json)
The reference's own module is imported from /root/reference and run on small YAML trees + command lines; what comes out
(the merged option tree) is the fixture.  easydict and termcolor are not installed: the stand-ins below give the reference
what it uses of them (attribute access on nested dicts; a colouring function that returns its text)."""
import io
import json
import os
import sys
import tempfile
import types
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class EasyDict(dict):
    """what the reference needs of easydict.EasyDict: nested dicts (also inside lists) become attribute-accessible"""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, cls):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


TREES = {
    "chain": {
        "root.yaml": "yaml:\nname: base\nseed: 0\ngpu: 0\ncpu:\noutput_root: output\n"
                     "data: {root: data, case: , image_size: [1920, 1080], frame_interval: 1}\n"
                     "PMVO: {patch_size: 9, conf_threshold: 0.1, threshold: 0.05, visible_threshold: 1, optimize: true,\n"
                     "       filter_point: true, infer_inner: true, genrate_ori_only: }\n"
                     "bbox_min: [-0.32, -0.32, -0.24]\nvsize: 0.005\n",
        "mid.yaml": "_parent_: {DIR}/root.yaml\nname: mid\nPMVO: {patch_size: 7, extra_mid: [1, 2, {a: 3}]}\ndata: {case: wavy}\n",
        "case.yaml": "_parent_: {DIR}/mid.yaml\ndata: {image_size: [1280, 720]}\nPMVO: {conf_threshold: 0.4}\n",
    },
    "two_parents": {
        "a.yaml": "yaml:\nname: a\nseed: 3\ngpu: 1\ncpu:\nsec: {x: 1, y: {z: 2}}\n",
        "b.yaml": "name: b\nsec: {x: 10, w: 5}\nother: text\n",
        "case.yaml": "_parent_: [{DIR}/a.yaml, {DIR}/b.yaml]\nsec: {y: {z: 20}}\n",
    },
}
RUNS = [
    ("chain", []),
    ("chain", ["--PMVO.infer_inner!", "--PMVO.optimize=", "--PMVO.filter_point", "--PMVO.threshold=0.025",
               "--data.image_size=[640,360]", "--name=run_7", "--vsize=5e-3", "--data.case=a.b"]),
    ("chain", ["--seed=3", "--gpu=2", "--PMVO.patch_size=4"]),
    ("chain", ["--seed=0", "--cpu", "--data.frame_interval=2", "--bbox_min=[-1,-2,-3.5]"]),
    ("two_parents", []),
    ("two_parents", ["--sec.y.z=7", "--other=null", "--sec.w=off", "--name=0123"]),
]


def main():
    sys.dont_write_bytecode = True
    sys.modules["easydict"] = types.SimpleNamespace(EasyDict=EasyDict)
    sys.modules["termcolor"] = types.SimpleNamespace(colored=lambda s, **k: str(s))
    sys.path.insert(0, "/root/reference")
    import options as ref          # the reference's module

    out = []
    for tree, argv in RUNS:
        with tempfile.TemporaryDirectory() as d:
            for fn, text in TREES[tree].items():
                open(os.path.join(d, fn), "w").write(text.replace("{DIR}", d))
            with redirect_stdout(io.StringIO()):
                cmd = ref.parse_arguments(["--yaml=%s" % os.path.join(d, "case")] + argv)
                opt = ref.set(cmd)
            res = ref.to_dict(opt)
            res.pop("yaml")
            out.append(dict(tree=tree, argv=argv, expect=res))
    json.dump(dict(trees=TREES, runs=out), open(os.path.join(ROOT, "tests", "golden", "options.json"), "w"), indent=1)
    for r in out:
        print(r["tree"], r["argv"], "->", r["expect"].get("name"), r["expect"].get("device"))


if __name__ == "__main__":
    main()
