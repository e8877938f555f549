// UNVERIFIED (no .NET SDK in the build image).  The compressor the reference declares only in a comment
// (csharp/src/ZStd.cs:89-96 Compress, :144-145 ZSTD_COMPRESSBOUND, :146-148 IsError / GetErrorName), over libzsmi.so.
using System;
using System.Runtime.InteropServices;

namespace EPAM.Deltix.ZStd
{
    public static unsafe class ZStdCompress
    {
        [DllImport(ZStdDecompress.Lib, CallingConvention = CallingConvention.Cdecl)]
        static extern UIntPtr zsmi_compress(void* dst, UIntPtr dstCapacity, void* src, UIntPtr srcSize, int level);
        [DllImport(ZStdDecompress.Lib, CallingConvention = CallingConvention.Cdecl)]
        static extern UIntPtr zsmi_compressBound(UIntPtr srcSize);
        [DllImport(ZStdDecompress.Lib, CallingConvention = CallingConvention.Cdecl)]
        static extern IntPtr zsmi_getErrorName(UIntPtr code);

        public const int DefaultLevel = 3;

        public static uint CompressBound(uint srcSize) => (uint)zsmi_compressBound((UIntPtr)srcSize);

        // one frame for the whole input; returns the frame size or (uint)(-code)
        public static uint Compress(byte[] dst, uint dstCapacity, byte[] src, uint srcSize, int compressionLevel = DefaultLevel)
        {
            if (dst == null || src == null) throw new ArgumentNullException(dst == null ? nameof(dst) : nameof(src));
            if (dstCapacity > (uint)dst.Length || srcSize > (uint)src.Length) throw new ArgumentOutOfRangeException();
            fixed (byte* d = dst, s = src)
                return unchecked((uint)(ulong)zsmi_compress(d, (UIntPtr)dstCapacity, s, (UIntPtr)srcSize, compressionLevel));
        }
        public static uint Compress(byte[] dst, byte[] src, int compressionLevel = DefaultLevel) => Compress(dst, (uint)dst.Length, src, (uint)src.Length, compressionLevel);

        public static byte[] Compress(byte[] src, int compressionLevel = DefaultLevel)
        {
            var dst = new byte[CompressBound((uint)src.Length)];
            uint r = Compress(dst, src, compressionLevel);
            if (ZStdDecompress.IsError(r)) throw new InvalidOperationException(GetErrorName(r));
            Array.Resize(ref dst, (int)r);
            return dst;
        }

        public static string GetErrorName(uint code) => Marshal.PtrToStringAnsi(zsmi_getErrorName((UIntPtr)(ulong)(long)(int)code));
    }
}
