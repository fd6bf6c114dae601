"""ORACLE package: CPU restatements of the reference's hot-path algorithm, used ONLY as the checker
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Never imported by the product
package (iad-r1_amd/); the product fails loudly when its HIP library is missing."""
