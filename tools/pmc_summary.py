"""Summarise a `rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv` run: per kernel, launches and mean counter value.
Usage: python tools/pmc_summary.py <dir with *counter_collection.csv> <out.csv>"""
import glob, os, sys
import pandas as pd

files = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
df = pd.concat([pd.read_csv(f) for f in files], ignore_index=True)
df["Kernel_Name"] = df["Kernel_Name"].str.slice(0, 120)
# one row per (dispatch, counter): sum over the instances rocprofv3 lists for a dispatch, then average over dispatches
per = df.groupby(["Kernel_Name", "Counter_Name", "Dispatch_Id"], as_index=False)["Counter_Value"].sum()
out = per.groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].agg(["count", "mean"]).reset_index()
out.to_csv(sys.argv[2], index=False)
print(out.to_string(index=False, max_colwidth=90))
