"""Dependency-free reader for the YAML subset the reference's input files use.

The reference reads its inputs with yaml-cpp (example/ecbs.cpp:554-574, example/sipp.cpp:181-205,
example/mapf_prioritized_sipp.cpp:189-209); the files themselves (benchmark/*/*.yaml, test/*.yaml) only ever use:

* block mappings      ``key: value`` / ``key:`` followed by an indented block
* block sequences     ``- item`` (the item may open a mapping: ``-   goal: [0, 6]`` with its other keys aligned below)
* flow sequences      ``[4, 21]``, ``[]``, nested ``[[0, 1], [2, 3]]``
* plain scalars       integers, floats, names such as ``agent0``; ``#`` comments; blank lines

That is what this module parses — into dicts / lists / ints / floats / strings — and nothing else: anchors, tags,
multi-line scalars, quoted strings with escapes and flow mappings raise ``ValueError``.
"""
from typing import Any, List, Tuple


def _scalar(tok: str) -> Any:
    t = tok.strip()
    if t == "" or t == "~" or t == "null":
        return None
    if len(t) >= 2 and t[0] == t[-1] and t[0] in "'\"":
        body = t[1:-1]
        if "\\" in body or t[0] in body:
            raise ValueError("escapes in quoted scalars are outside the supported subset: %r" % tok)
        return body
    if t in ("true", "True"):
        return True
    if t in ("false", "False"):
        return False
    try:
        return int(t)
    except ValueError:
        pass
    try:
        return float(t)
    except ValueError:
        pass
    if t[0] in "&*!|>{%@`":
        raise ValueError("outside the supported YAML subset: %r" % tok)
    return t


def _flow(text: str, pos: int) -> Tuple[Any, int]:
    """Parse a flow sequence starting at text[pos] == '['; returns (list, index after the closing bracket)."""
    assert text[pos] == "["
    out: List[Any] = []
    pos += 1
    tok = ""
    expect_item = False
    while True:
        if pos >= len(text):
            raise ValueError("unterminated flow sequence: %r" % text)
        c = text[pos]
        if c == "[":
            if tok.strip():
                raise ValueError("unexpected '[' in %r" % text)
            item, pos = _flow(text, pos)
            out.append(item)
            tok = ""
            expect_item = False
            # skip to the next ',' or ']'
            while pos < len(text) and text[pos] == " ":
                pos += 1
            if pos < len(text) and text[pos] == ",":
                pos += 1
                expect_item = True
            continue
        if c == "]":
            if tok.strip():
                out.append(_scalar(tok))
            elif expect_item:
                raise ValueError("dangling ',' in %r" % text)
            return out, pos + 1
        if c == ",":
            if not tok.strip():
                raise ValueError("empty item in %r" % text)
            out.append(_scalar(tok))
            tok = ""
            expect_item = True
            pos += 1
            continue
        if c in "{}":
            raise ValueError("flow mappings are outside the supported subset: %r" % text)
        tok += c
        pos += 1


def _value(text: str) -> Any:
    t = text.strip()
    if t.startswith("["):
        v, end = _flow(t, 0)
        if t[end:].strip():
            raise ValueError("trailing characters after flow sequence: %r" % text)
        return v
    return _scalar(t)


def _strip_comment(line: str) -> str:
    # '#' starts a comment at line start or after whitespace (the inputs never quote a '#')
    for i, c in enumerate(line):
        if c == "#" and (i == 0 or line[i - 1] in " \t"):
            return line[:i]
    return line


def loads(text: str) -> Any:
    lines: List[Tuple[int, str]] = []
    for raw in text.splitlines():
        if "\t" in raw[:len(raw) - len(raw.lstrip())]:
            raise ValueError("tabs in indentation")
        s = _strip_comment(raw).rstrip()
        if not s.strip() or s.strip() == "---":
            continue
        lines.append((len(s) - len(s.lstrip(" ")), s.strip()))
    if not lines:
        return None
    node, nxt = _block(lines, 0, lines[0][0])
    if nxt != len(lines):
        raise ValueError("unexpected dedent / content at line %d: %r" % (nxt, lines[nxt][1]))
    return node


def _split_key(s: str) -> Tuple[str, str]:
    """'key: rest' -> (key, rest); raises if the line is not a mapping entry."""
    i = s.find(":")
    while i != -1 and not (i + 1 == len(s) or s[i + 1] == " "):
        i = s.find(":", i + 1)
    if i <= 0:
        raise ValueError("expected 'key: value', got %r" % s)
    return s[:i].strip(), s[i + 1:].strip()


def _is_key_line(s: str) -> bool:
    if s.startswith("[") or s.startswith("- ") or s == "-":
        return False
    i = s.find(":")
    while i != -1 and not (i + 1 == len(s) or s[i + 1] == " "):
        i = s.find(":", i + 1)
    return i > 0


def _block(lines: List[Tuple[int, str]], i: int, indent: int) -> Tuple[Any, int]:
    """Parse the block whose first line is lines[i] at column `indent`."""
    ind, s = lines[i]
    if ind != indent:
        raise ValueError("bad indentation at %r" % s)
    if s.startswith("- ") or s == "-":
        seq: List[Any] = []
        while i < len(lines) and lines[i][0] == indent and (lines[i][1].startswith("- ") or lines[i][1] == "-"):
            body = lines[i][1][1:]
            pad = len(body) - len(body.lstrip(" "))
            body = body.strip()
            if body == "":
                if i + 1 < len(lines) and lines[i + 1][0] > indent:
                    item, i = _block(lines, i + 1, lines[i + 1][0])
                else:
                    item, i = None, i + 1
            elif _is_key_line(body) or body.startswith("- ") or body == "-":
                # "- key: value" opens a mapping whose further keys are aligned with `key`; "- - x" a nested sequence
                col = indent + 1 + pad
                sub = [(col, body)]
                j = i + 1
                while j < len(lines) and lines[j][0] >= col:
                    sub.append(lines[j])
                    j += 1
                item, used = _block(sub, 0, col)
                if used != len(sub):
                    raise ValueError("bad indentation inside sequence item near %r" % sub[used][1])
                i = j
            else:
                item, i = _value(body), i + 1
            seq.append(item)
        if i < len(lines) and lines[i][0] > indent:
            raise ValueError("bad indentation at %r" % lines[i][1])
        return seq, i
    if _is_key_line(s):
        mp = {}
        while i < len(lines) and lines[i][0] == indent and _is_key_line(lines[i][1]):
            key, rest = _split_key(lines[i][1])
            if key in mp:
                raise ValueError("duplicate key %r" % key)
            if rest != "":
                mp[key] = _value(rest)
                i += 1
            elif i + 1 < len(lines) and (lines[i + 1][0] > indent or
                                         (lines[i + 1][0] == indent and lines[i + 1][1].startswith("-"))):
                # a block sequence may sit at the same column as its key (PyYAML's default dump style)
                mp[key], i = _block(lines, i + 1, lines[i + 1][0])
            else:
                mp[key] = None
                i += 1
        if i < len(lines) and lines[i][0] > indent:
            raise ValueError("bad indentation at %r" % lines[i][1])
        return mp, i
    return _value(s), i + 1


def load(path: str) -> Any:
    with open(path) as f:
        return loads(f.read())
