import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["AverageNs"]) > 20000 and int(r["Calls"]) > 5:
        print("   %-72s calls %4s avg %9.1f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"])/1e3))
