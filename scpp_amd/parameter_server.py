"""Python mirror of the reference ParameterServer (scpp_core/utils/include/parameterServer.hpp:34-127):
typed scalar / vector loading from Boost-INFO files (`key value`, `key { (i) value ... }`, `;` comments)."""
import re


class ParameterServer:
    def __init__(self, filename):
        with open(filename) as f:
            text = f.read()
        toks = []
        for ln, line in enumerate(text.splitlines()):
            line = line.split(";", 1)[0]
            for m in re.finditer(r"[{}]|[^\s{}]+", line):
                toks.append((m.group(0), ln))
        self.tree = {}
        stack = [self.tree]
        i = 0
        while i < len(toks):
            tok, ln = toks[i]
            if tok == "}":
                if len(stack) < 2:
                    raise ValueError(f"INFO parse error: unmatched }} in {filename}")
                stack.pop()
                i += 1
                continue
            if tok == "{":
                raise ValueError(f"INFO parse error: unexpected {{ in {filename}")
            node = {"value": None, "children": {}}
            stack[-1][tok] = node
            i += 1
            if i < len(toks) and toks[i][1] == ln and toks[i][0] not in "{}":
                node["value"] = toks[i][0]
                i += 1
            if i < len(toks) and toks[i][0] == "{":
                stack.append(node["children"])
                i += 1

    def _get(self, name):
        if name not in self.tree or self.tree[name]["value"] is None:
            raise RuntimeError(f"WARNING: Failed to load scalar type: {name}!")
        return self.tree[name]["value"]

    def load_scalar(self, name, typ=float):
        v = self._get(name)
        if typ is bool:
            if v in ("true", "1"):
                return True
            if v in ("false", "0"):
                return False
            raise RuntimeError(f"WARNING: Failed to load scalar type: {name}!")
        try:
            return typ(float(v)) if typ is int else typ(v)
        except ValueError:
            raise RuntimeError(f"WARNING: Failed to load scalar type: {name}!")

    def load_vector(self, name, rows):
        if name not in self.tree:
            raise RuntimeError(f"Failed to load matrix type: {name}!")
        ch = self.tree[name]["children"]
        scaling = float(ch["scaling"]["value"]) if "scaling" in ch else 1.0
        entries = len(ch) - (1 if "scaling" in ch else 0)
        if entries < rows:
            raise RuntimeError(f"Missing entries in matrix type: {name}!")
        if entries > rows:
            raise RuntimeError(f"Redundant entries in matrix type: {name}!")
        out = []
        for i in range(rows):
            key = f"({i})"
            if key not in ch:
                raise RuntimeError(f"Failed to load matrix type: {name}!")
            out.append(float(ch[key]["value"]) * scaling)
        return out
