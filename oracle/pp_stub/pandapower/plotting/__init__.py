"""import-time placeholder for pandapower.plotting (pf_res_plot.py:9-13 imports names from it; never called)"""
