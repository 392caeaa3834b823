for f in scratch/lib_np*.so; do echo "== $f"; ESAC_HIP_LIB=$PWD/$f python scratch/cyc.py 2>/dev/null | grep -E "total|error_images|point_loop|per pass"; done
