"""INTEGRATION.md section 4's option table from juqbox.jl_amd/csrc/jq_options.h (the one place options are defined): rewrites the lines
between <!-- options:begin --> and <!-- options:end -->.  tests/test_abi.py checks that every option has its row."""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "juqbox.jl_amd", "csrc", "jq_options.h")).read()
rows = re.findall(r'^    \{"([a-z0-9_]+)", (JQ_OPT_UNSET|-?\d+), ([A-Z_| 0]+), "(.*)"\},$', src, flags=re.M)
out = ["| option | default | kind | effect |", "|---|---|---|---|"]
for name, dflt, flags, doc in rows:
    kind = ", ".join(k for k, f in (("plan", "JQ_OPT_PLAN"), ("test hook", "JQ_OPT_HOOK"), ("experiment builds only", "JQ_OPT_EXP")) if f in flags) or "per evaluation"
    out.append("| `%s` | %s | %s | %s |" % (name, "not set" if dflt == "JQ_OPT_UNSET" else dflt, kind, doc.replace("|", "\\|")))
p = os.path.join(ROOT, "INTEGRATION.md")
txt = open(p).read()
a, b = txt.index("<!-- options:begin -->"), txt.index("<!-- options:end -->")
open(p, "w").write(txt[:a] + "<!-- options:begin -->\n" + "\n".join(out) + "\n" + txt[b:])
print("%d options" % len(rows))
