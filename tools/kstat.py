import sqlite3,sys
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
v=[t for t in tabs if t=='kernels' or t.startswith('kernels')][0]
rows=cur.execute(f"select name, count(*), avg(end-start)/1000.0, sum(end-start)/1000.0 from {v} group by name order by 4 desc").fetchall()
tot=sum(r[3] for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 16]: print(f"{r[0][:80]:80s} {r[1]:4d} {r[2]:9.1f} {100*r[3]/tot:5.1f}%")
