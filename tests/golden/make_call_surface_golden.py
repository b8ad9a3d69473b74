"""Generates tests/golden/call_surface.json: the NAMES the reference uses at the boundary of the path — no source text.

Run in the build container (the reference does not travel):  python tests/golden/make_call_surface_golden.py
Parsed with `ast` from /root/reference:
  gaussian_renderer/__init__.py   keywords of the GaussianRasterizationSettings(...) call (:37-53), keywords of the rasterizer(...)
                                  call (:95-107), the names the 5-tuple is unpacked into (:94), render()'s parameters and defaults
                                  (:18), the keys of the dict it returns (:112-119), the model / camera / pipe attributes it reads
  train.py, render.py, viewer.py, render_traj.py
                                  keywords of every render(...) call, every key read from a render result
"""
import ast
import json
import os

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "call_surface.json")


def parse(rel):
    with open(os.path.join(REF, rel)) as f:
        return ast.parse(f.read())


def call_name(node):
    f = node.func
    return f.id if isinstance(f, ast.Name) else (f.attr if isinstance(f, ast.Attribute) else None)


def const(node):
    try:
        return ast.literal_eval(node)
    except Exception:
        return ast.unparse(node)


def main():
    out = {}
    tree = parse("gaussian_renderer/__init__.py")
    render_def = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "render")
    a = render_def.args
    names = [x.arg for x in a.args]
    defaults = [const(d) for d in a.defaults]
    out["render_parameters"] = names
    out["render_defaults"] = dict(zip(names[len(names) - len(defaults):], defaults))
    for node in ast.walk(render_def):
        if isinstance(node, ast.Call) and call_name(node) == "GaussianRasterizationSettings":
            out["settings_keywords"] = [k.arg for k in node.keywords]
            assert not node.args
        if isinstance(node, ast.Assign) and isinstance(node.value, ast.Call) and call_name(node.value) == "rasterizer":
            out["rasterizer_keywords"] = [k.arg for k in node.value.keywords]
            assert not node.value.args
            tgt = node.targets[0]
            out["rasterizer_returns"] = [ast.unparse(e) for e in tgt.elts]
        if isinstance(node, ast.Call) and call_name(node) == "GaussianRasterizer":
            out["rasterizer_constructor_keywords"] = [k.arg for k in node.keywords]
        if isinstance(node, ast.Return) and isinstance(node.value, ast.Dict):
            out["render_result_keys"] = [const(k) for k in node.value.keys]
    attrs = {"pc": set(), "viewpoint_camera": set(), "pipe": set()}
    for node in ast.walk(render_def):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id in attrs:
            attrs[node.value.id].add(node.attr)
    out["attributes_read"] = {k: sorted(v) for k, v in attrs.items()}

    callers = {}
    for rel in ("train.py", "render.py", "viewer.py", "render_traj.py"):
        t = parse(rel)
        kws, keys = set(), set()
        result_names = set()
        for node in ast.walk(t):
            if isinstance(node, ast.Call) and call_name(node) == "render":
                kws.update(k.arg for k in node.keywords if k.arg)
                callers.setdefault("max_positional_arguments", 0)
                callers["max_positional_arguments"] = max(callers["max_positional_arguments"], len(node.args))
            if isinstance(node, ast.Assign) and isinstance(node.value, ast.Call) and call_name(node.value) == "render":
                for tg in node.targets:
                    if isinstance(tg, ast.Name):
                        result_names.add(tg.id)
        for node in ast.walk(t):
            if isinstance(node, ast.Subscript) and isinstance(node.slice, ast.Constant) and isinstance(node.slice.value, str):
                v = node.value
                if (isinstance(v, ast.Name) and v.id in result_names) or (isinstance(v, ast.Call) and call_name(v) == "render"):
                    keys.add(node.slice.value)
        callers[rel] = {"render_keywords": sorted(kws), "result_keys_read": sorted(keys)}
    out["callers"] = callers
    out["import_line_names"] = sorted(
        al.name for n in ast.walk(tree) if isinstance(n, ast.ImportFrom) and n.module == "diff_gaussian_rasterization"
        for al in n.names)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
