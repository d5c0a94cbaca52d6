import csv, collections, sys
c=collections.Counter(); names=collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    k=(r["Queue_Id"],r["Stream_Id"])
    c[k]+=1
    names[k][r["Kernel_Name"].replace("(anonymous namespace)::","")[:34]]+=1
for k,v in sorted(c.items()):
    print(k,v,names[k].most_common(2))
