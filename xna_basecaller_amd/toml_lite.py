"""
Minimal TOML reader/writer for model `config.toml` files (Python 3.10 has no tomllib and the
`toml` wheel the reference uses -- ub-bonito/bonito/util.py:16,277 -- is not installed here).

Supports exactly what the reference's configs contain (models/*/config.toml, and the
`[training]`/`[basecaller]` tables bonito train appends, cli/train.py:113-114): tables
`[a]` / `[a.b]`, `key = value` with strings, integers, floats, booleans and (possibly
multi-line, trailing-comma) arrays of those, `#` comments.
"""
import re

__all__ = ["loads", "load", "dumps"]

_num = re.compile(r"^[+-]?(\d[\d_]*)(\.\d[\d_]*)?([eE][+-]?\d+)?$")


def _strip_comment(line):
    out, q = [], None
    for i, ch in enumerate(line):
        if q:
            if ch == q and line[i - 1] != "\\":
                q = None
        elif ch in "\"'":
            q = ch
        elif ch == "#":
            break
        out.append(ch)
    return "".join(out).strip()


def _scalar(tok):
    tok = tok.strip()
    if not tok:
        raise ValueError("empty TOML value")
    if tok[0] == '"' and tok[-1] == '"':
        return bytes(tok[1:-1], "utf-8").decode("unicode_escape")
    if tok[0] == "'" and tok[-1] == "'":
        return tok[1:-1]
    if tok == "true":
        return True
    if tok == "false":
        return False
    if tok in ("inf", "+inf"):
        return float("inf")
    if tok == "-inf":
        return float("-inf")
    if tok == "nan":
        return float("nan")
    if _num.match(tok):
        t = tok.replace("_", "")
        return float(t) if any(c in t for c in ".eE") else int(t)
    raise ValueError("unsupported TOML value: %r" % tok)


def _split_array(body):
    items, depth, q, cur = [], 0, None, []
    for i, ch in enumerate(body):
        if q:
            cur.append(ch)
            if ch == q and body[i - 1] != "\\":
                q = None
            continue
        if ch in "\"'":
            q = ch
            cur.append(ch)
        elif ch == "[":
            depth += 1
            cur.append(ch)
        elif ch == "]":
            depth -= 1
            cur.append(ch)
        elif ch == "," and depth == 0:
            items.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    items.append("".join(cur))
    return [x.strip() for x in items if x.strip()]


def _value(tok):
    tok = tok.strip()
    if tok.startswith("["):
        if not tok.endswith("]"):
            raise ValueError("unterminated TOML array: %r" % tok)
        return [_value(x) for x in _split_array(tok[1:-1])]
    return _scalar(tok)


def _balanced(s):
    depth, q = 0, None
    for i, ch in enumerate(s):
        if q:
            if ch == q and s[i - 1] != "\\":
                q = None
        elif ch in "\"'":
            q = ch
        elif ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
    return depth == 0


def loads(text):
    root = {}
    table = root
    pending = None
    for raw in text.splitlines():
        line = _strip_comment(raw)
        if pending is not None:
            pending[1] += " " + line
            if _balanced(pending[1]):
                table[pending[0]] = _value(pending[1])
                pending = None
            continue
        if not line:
            continue
        if line.startswith("[") and "=" not in line.split("]")[0]:
            is_array = line.startswith("[[")              # [[a.b]]: append a new table to the array a.b
            parts = [p.strip().strip('"') for p in line.strip("[] \t").split(".")]
            table = root
            for i, part in enumerate(parts):
                last = i == len(parts) - 1
                if last and is_array:
                    arr = table.setdefault(part, [])
                    arr.append({})
                    table = arr[-1]
                else:
                    nxt = table.setdefault(part, {})
                    table = nxt[-1] if isinstance(nxt, list) else nxt       # [a.b.c] below [[a.b]]: its latest element
            continue
        key, sep, val = line.partition("=")
        if not sep:
            raise ValueError("bad TOML line: %r" % raw)
        key = key.strip().strip('"')
        val = val.strip()
        if val.startswith("[") and not _balanced(val):
            pending = [key, val]
        else:
            table[key] = _value(val)
    if pending is not None:
        raise ValueError("unterminated TOML array for key %r" % pending[0])
    return root


def load(path):
    with open(path, "r", encoding="utf-8") as fh:
        return loads(fh.read())


def _fmt(v):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, (int, float)):
        return repr(v)
    if isinstance(v, str):
        return '"%s"' % v.replace("\\", "\\\\").replace('"', '\\"')
    if isinstance(v, (list, tuple)):
        return "[ " + ", ".join(_fmt(x) for x in v) + ",]" if v else "[]"
    raise TypeError("cannot encode %r" % (v,))


def dumps(d, _prefix=""):
    lines, tables, arrays = [], [], []
    for k, v in d.items():
        if isinstance(v, dict):
            tables.append((k, v))
        elif isinstance(v, (list, tuple)) and v and all(isinstance(x, dict) for x in v):
            arrays.append((k, v))                         # array of tables: [[prefix.k]] per element
        else:
            lines.append("%s = %s" % (k, _fmt(v)))
    out = "\n".join(lines)
    for k, v in tables:
        name = _prefix + k
        body = dumps(v, name + ".")
        out += ("\n\n" if out else "") + "[%s]\n%s" % (name, body)
    for k, v in arrays:
        name = _prefix + k
        for item in v:
            out += ("\n\n" if out else "") + "[[%s]]\n%s" % (name, dumps(item, name + "."))
    return out + ("\n" if not _prefix and not out.endswith("\n") else "")
