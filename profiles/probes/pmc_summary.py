"""Per-kernel mean of a PMC counter from rocprofv3 rocpd databases (one database per counter pass).
usage: pmc_summary.py FETCH_SIZE=fetch.db WRITE_SIZE=write.db ... [--json out.json --batch B]"""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    return re.sub(r"<(\d+), (false|true)>", r"<\1>", name)


def load(db):
    c = sqlite3.connect(db)
    out = {}
    for name, cname, val in c.execute("select name, counter_name, counter_value from pmc_events"):
        out.setdefault((short(name), cname), []).append(val)
    return out


def main():
    args = [a for a in sys.argv[1:] if "=" in a]
    data = {}
    for a in args:
        cname, db = a.split("=", 1)
        for (k, cn), vals in load(db).items():
            if not k.startswith("k_"):
                continue
            data.setdefault(k, {})[cn] = (sum(vals) / len(vals), len(vals))
    for k in sorted(data):
        print(k, {cn: ("%.1f" % v[0], v[1]) for cn, v in data[k].items()})
    return data


if __name__ == "__main__":
    main()
