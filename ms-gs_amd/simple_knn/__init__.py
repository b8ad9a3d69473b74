"""MI355X stand-in for the reference's `simple_knn` submodule (un-vendored; /root/reference/.gitmodules).
`from simple_knn._C import distCUDA2` (/root/reference/scene/gaussian_model.py:26) resolves here when ms-gs_amd/ is
on sys.path."""
