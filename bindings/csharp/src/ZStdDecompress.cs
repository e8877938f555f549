// UNVERIFIED (no .NET SDK in the build image).  Drop-in for the reference's public static class
// EPAM.Deltix.ZStd.ZStdDecompress (csharp/src/ZStdDecompress.cs:37-42): same four public signatures, same return
// conventions (sizes are 32-bit; an error is (uint)(-code) with the codes of ZStdErrors.cs:61-90, never an exception),
// bodies replaced by P/Invoke into libzsmi.so (include/zsmi.h).  The GPU does the decoding; nothing is retained past return.
using System;
using System.Runtime.InteropServices;

namespace EPAM.Deltix.ZStd
{
    public static unsafe class ZStdDecompress
    {
        internal const string Lib = "zsmi";   // libzsmi.so / zsmi.dll on the loader path

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        static extern UIntPtr zsmi_decompress(void* dst, UIntPtr dstCapacity, void* src, UIntPtr srcSize);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        static extern ulong zsmi_getDecompressedSize(void* src, UIntPtr srcSize);

        // replaces ZStdDecompress.cs:2182-2186
        public static uint Decompress(byte[] dst, uint dstCapacity, byte[] src, uint srcSize)
        {
            if (dst == null || src == null) throw new ArgumentNullException(dst == null ? nameof(dst) : nameof(src));
            if (dstCapacity > (uint)dst.Length || srcSize > (uint)src.Length) throw new ArgumentOutOfRangeException();
            fixed (byte* d = dst, s = src)
                return unchecked((uint)(ulong)zsmi_decompress(d, (UIntPtr)dstCapacity, s, (UIntPtr)srcSize));   // (size_t)-code truncates to (uint)-code
        }

        // replaces ZStdDecompress.cs:2188-2191
        public static uint Decompress(byte[] dst, byte[] src) => Decompress(dst, (uint)dst.Length, src, (uint)src.Length);

        // replaces ZStdDecompress.cs:590-601
        public static ulong GetDecompressedSize(byte[] src) => GetDecompressedSize(src, (uint)src.Length);

        // replaces ZStdDecompress.cs:603-622: 0 for unknown / error / skippable frame
        public static ulong GetDecompressedSize(byte[] src, uint srcSize)
        {
            if (src == null) throw new ArgumentNullException(nameof(src));
            if (srcSize > (uint)src.Length) throw new ArgumentOutOfRangeException(nameof(srcSize));
            fixed (byte* s = src) return zsmi_getDecompressedSize(s, (UIntPtr)srcSize);
        }

        // ZStdErrors.IsError is internal in the reference (ZStdErrors.cs:95-98); callers compare with this bound
        public static bool IsError(uint code) => code > unchecked((uint)-120);
    }
}
