"""The argument namespaces the reference's CLIs hand to ``build_model``, as a data fixture (build container only):
    python tools/gen_golden_args.py   ->   tests/golden/args.json

``get_args_parser()`` of main.py (:31-193) and main_multi.py (:28-177) cannot be imported (the modules pull datasets,
engine, wandb ...), so the function's source lines are cut out of the reference file with ``ast`` at generation time and
executed here (argparse only) - nothing of it is stored.  The fixture holds, per script, the parser's defaults
(``parse_args([])``) and, per shipped ``configs/training/*.sh``, the flags its ``python -u main*.py`` command passes
(shell variables substituted by the values the script assigns them) with the namespace the parser makes of them.
tests/test_args_contract.py builds every configuration from these namespaces and holds models/config.py to them."""
import argparse
import ast
import glob
import json
import os
import re
import shlex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def parser_of(script):
    src = open(os.path.join(REF, script)).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "get_args_parser")
    code = "\n".join(src.splitlines()[fn.lineno - 1:fn.end_lineno])
    import numpy
    scope = {"argparse": argparse, "np": numpy}
    exec(compile(code, script, "exec"), scope)      # the reference's own function, executed - not stored
    return scope["get_args_parser"]()


def plain(ns):
    return {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in sorted(vars(ns).items())}


def command_of(path):
    """(script, flags) of the `python -u main*.py ...` command of a config script."""
    text = open(path).read()
    variables = {}
    for m in re.finditer(r"^([A-Z_]+)\s*=\s*([^\s#]+)", text, flags=re.M):
        variables[m.group(1)] = m.group(2)
    body = text[text.index("python -u"):].replace("\\\n", " ")
    body = body.split("|")[0]
    body = re.sub(r"\$\{([A-Z_]+)\}", lambda m: variables.get(m.group(1), m.group(0)), body)
    tokens = shlex.split(body)
    return tokens[2], tokens[3:]


doc = {"parsers": {}, "configs": {}}
parsers = {s: parser_of(s) for s in ("main.py", "main_multi.py")}
for script, p in parsers.items():
    doc["parsers"][script] = plain(p.parse_args([]))
for path in sorted(glob.glob(os.path.join(REF, "configs", "training", "*.sh"))):
    script, flags = command_of(path)
    doc["configs"][os.path.basename(path)] = {"script": script, "flags": flags, "namespace": plain(parsers[script].parse_args(flags))}
out = os.path.join(ROOT, "tests", "golden", "args.json")
with open(out, "w") as fh:
    json.dump(doc, fh, indent=1, sort_keys=True)
print("wrote", out, {k: len(v["flags"]) for k, v in doc["configs"].items()})
