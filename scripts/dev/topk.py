import sqlite3,sys
db=sqlite3.connect(sys.argv[1])
rows=list(db.execute("select * from top_kernels"))
tot=sum(r[2] for r in rows)
print("total us", tot)
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 14]:
    print(str(r[0])[:100], r[1], round(r[2]/1e3,1),"ms", round(r[3],1), round(r[4],2))
