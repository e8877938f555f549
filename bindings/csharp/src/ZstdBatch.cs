// UNVERIFIED (no .NET SDK in the build image).  The data-parallel hot path: n independent chunks <-> n frames in one call
// (zsmi_compressBatchHost / zsmi_decompressBatchHost of include/zsmi.h; the device-pointer forms are for native hosts).
using System;
using System.Runtime.InteropServices;

namespace EPAM.Deltix.ZStd
{
    public sealed unsafe class ZstdBatch : IDisposable
    {
        [DllImport(ZStdDecompress.Lib, CallingConvention = CallingConvention.Cdecl)] static extern IntPtr zsmi_createCtx(int device, IntPtr hipStream);
        [DllImport(ZStdDecompress.Lib, CallingConvention = CallingConvention.Cdecl)] static extern void zsmi_freeCtx(IntPtr ctx);
        [DllImport(ZStdDecompress.Lib, CallingConvention = CallingConvention.Cdecl)]
        static extern int zsmi_compressBatchHost(IntPtr ctx, void* src, ulong* srcOffsets, uint* srcSizes, uint n, void* dst, ulong* dstOffsets, uint* dstSizes, int level);
        [DllImport(ZStdDecompress.Lib, CallingConvention = CallingConvention.Cdecl)]
        static extern int zsmi_decompressBatchHost(IntPtr ctx, void* src, ulong* srcOffsets, uint* srcSizes, uint n, void* dst, ulong* dstOffsets, uint* dstCaps, uint* dstSizes);

        IntPtr ctx;
        public ZstdBatch(int device = -1)
        {
            ctx = zsmi_createCtx(device, IntPtr.Zero);
            if (ctx == IntPtr.Zero) throw new InvalidOperationException("no usable HIP device (this codec has no CPU path)");
        }
        public void Dispose() { if (ctx != IntPtr.Zero) { zsmi_freeCtx(ctx); ctx = IntPtr.Zero; } }

        // dstSizes[i] = frame bytes, or (uint)(-code) for a chunk that failed; dstOffsets[i] must leave CompressBound(srcSizes[i]) bytes
        public void Compress(byte[] src, ulong[] srcOffsets, uint[] srcSizes, byte[] dst, ulong[] dstOffsets, uint[] dstSizes, int level = 3)
        {
            fixed (byte* s = src, d = dst) fixed (ulong* so = srcOffsets, dof = dstOffsets) fixed (uint* ss = srcSizes, ds = dstSizes)
            { int rc = zsmi_compressBatchHost(ctx, s, so, ss, (uint)srcSizes.Length, d, dof, ds, level); if (rc != 0) throw new InvalidOperationException("zsmi error " + rc); }
        }
        public void Decompress(byte[] src, ulong[] srcOffsets, uint[] srcSizes, byte[] dst, ulong[] dstOffsets, uint[] dstCaps, uint[] dstSizes)
        {
            fixed (byte* s = src, d = dst) fixed (ulong* so = srcOffsets, dof = dstOffsets) fixed (uint* ss = srcSizes, dc = dstCaps, ds = dstSizes)
            { int rc = zsmi_decompressBatchHost(ctx, s, so, ss, (uint)srcSizes.Length, d, dof, dc, ds); if (rc != 0) throw new InvalidOperationException("zsmi error " + rc); }
        }
    }
}
