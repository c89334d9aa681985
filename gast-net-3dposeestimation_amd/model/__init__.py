"""Drop-in replacement of the reference's `model` package (reference model/gast_net.py, local_attention.py,
global_attention.py, sem_graph_conv.py).  Put the directory that contains this package first on sys.path and the
reference's trainval.py / reconstruction.py / gen_skes.py import it unchanged."""
