import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'k_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows)//2: len(rows)//2 + 60]
t0 = int(rows[0]['Start_Timestamp'])
import re
for r in rows:
    m = re.search(r'(k_[a-z_]+)', r['Kernel_Name'])
    print(r.get('Queue_Id'), r.get('Stream_Id', ''), m.group(1), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3)
