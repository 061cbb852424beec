import csv,glob,statistics,sys
for v in sys.argv[1:]:
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        vals={}
        for f in glob.glob(f"gpurun_out/r04/ex_{v}{c}/**/*counter_collection.csv",recursive=True):
            for row in csv.DictReader(open(f)):
                if "bf_top2" in row["Kernel_Name"]:
                    vals.setdefault(row["Dispatch_Id"],0.0); vals[row["Dispatch_Id"]]+=float(row["Counter_Value"])
        print(v,c,"median KB per launch", statistics.median(vals.values()) if vals else None)
