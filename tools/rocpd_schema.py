#!/usr/bin/env python
"""Print the tables / views (+ columns) of a rocprofv3 rocpd sqlite database."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
for name, typ in con.execute("select name, type from sqlite_master where type in ('table','view') order by type, name"):
    cols = [r[1] for r in con.execute(f"pragma table_info('{name}')")]
    n = con.execute(f"select count(*) from '{name}'").fetchone()[0]
    print(typ, name, n, cols)
