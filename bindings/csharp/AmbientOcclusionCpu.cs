// AmbientOcclusionCpu.cs -- scalar C# CPU implementation of the same passes (north_star: "a scalar
// C# CPU reimplementation of the same passes"), mechanically parallel to oracle/meao_oracle.c
// (gather form: every texel a pure function of the depth image).
//
// NOT COMPILED OR TIMED HERE: the build image has no dotnet/mono/csc.  It IS executed, though:
// tests/test_csharp_cpu_mirror.py runs this file through oracle/csharp_interp.py on small frames
// and requires all 17 buffers to equal the C oracle's bit for bit.  Written against modern .NET
// (MathF.FusedMultiplyAdd, BitConverter.SingleToInt32Bits); reference storage formats only
// (R8 AO, f16 depth with round-toward-zero / clamp-to-65504, 4 levels, f32 input depth).
//
// Citations: AO.cs = Assets/MiniEngineAO/AmbientOcclusion.cs, DS1/DS2/REN/UPS =
// Assets/MiniEngineAO/Shaders/{Downsample1,Downsample2,Render,Upsample}.compute of the reference.

using System;

namespace MiniEngineAO.Cpu
{
    public sealed class AmbientOcclusionCpu
    {
        // component properties (AO.cs:20-68) and camera terms (AO.cs:561-573)
        public float noiseFilterTolerance = 0;
        public float blurTolerance = -4.6f;
        public float upsampleTolerance = -12;
        public float thicknessModifier = 1;
        public float intensity = 1;
        public float nearClipPlane = 0.3f;
        public float farClipPlane = 1000;
        public float projection00 = 0.9742786f;
        public bool usesReversedZBuffer = true;

        // the 17 debug-visible buffers (AO.cs:453-475)
        public int width;
        public int height;
        public ushort[] linearDepth;
        public float[][] lowDepth = new float[4][];
        public ushort[][] tiledDepth = new ushort[4][];
        public byte[][] occlusion = new byte[4][];
        public byte[][] combined = new byte[3][];
        public byte[] result;

        float zp0;
        float zp1;
        float[] depth;

        // ---- helpers -----------------------------------------------------------------------
        static int LevelDim(int v, int level)              // AO.cs:276-281
        {
            int div = 1 << level;
            return (v + (div - 1)) / div;
        }

        static int ClampI(int v, int lo, int hi)
        {
            return v < lo ? lo : (v > hi ? hi : v);
        }

        static float Saturate(float x)                      // D3D: NaN -> 0
        {
            if (!(x == x)) return 0.0f;
            return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
        }

        static float ClampF(float x, float lo, float hi)    // min(max(x, lo), hi), NaN x -> lo
        {
            float m = (x == x && x > lo) ? x : lo;
            return m < hi ? m : hi;
        }

        static float Mad(float a, float b, float c)
        {
            return MathF.FusedMultiplyAdd(a, b, c);
        }

        // f32 -> f16, round toward zero, finite overflow clamps to 65504 (HalfUAV store)
        static ushort ToHalfRtz(float x)
        {
            int u = BitConverter.SingleToInt32Bits(x);
            int sign = (u >> 16) & 0x8000;
            int absu = u & 0x7fffffff;
            if (absu >= 0x7f800000)
                return (ushort)(sign | (absu == 0x7f800000 ? 0x7c00 : (0x7e00 | ((absu >> 13) & 0x1ff))));
            int e = (absu >> 23) - 127;
            int m = absu & 0x7fffff;
            if (e > 15) return (ushort)(sign | 0x7bff);
            if (e >= -14) return (ushort)(sign | ((e + 15) << 10) | (m >> 13));
            if (e >= -25) return (ushort)(sign | ((m | 0x800000) >> (-e - 1)));
            return (ushort)sign;
        }

        static float FromHalf(ushort hv)
        {
            int h = hv;
            int sign = (h & 0x8000) << 16;
            int e = (h >> 10) & 0x1f;
            int m = h & 0x3ff;
            if (e == 31) return BitConverter.Int32BitsToSingle(sign | 0x7f800000 | (m << 13));
            if (e == 0)
            {
                float v = (float)m * 5.9604644775390625e-8f;
                return sign != 0 ? -v : v;
            }
            return BitConverter.Int32BitsToSingle(sign | ((e + 112) << 23) | (m << 13));
        }

        static byte ToUnorm8(float x)                       // FixedUAV store
        {
            float s = Saturate(x) * 255.0f;
            s = s + 0.5f;
            return (byte)(int)s;
        }

        static float FromUnorm8(byte v)
        {
            return (float)v / 255.0f;
        }

        // ---- pass 1+2: linearize, point-downsample, de-interleave (DS1, DS2) -------------------
        float Linearize(int x, int y)                       // DS1:37-48
        {
            float d = (x < width && y < height) ? depth[y * width + x] : 0.0f;
            float dist = 1.0f / Mad(zp0, d, zp1);
            if (usesReversedZBuffer ? (d == 0.0f) : (d == 1.0f)) dist = 1e5f;
            return dist;
        }

        void Downsample()
        {
            linearDepth = new ushort[width * height];
            for (int y = 0; y < height; y++)
                for (int x = 0; x < width; x++)
                    linearDepth[y * width + x] = ToHalfRtz(Linearize(x, y));
            for (int k = 1; k <= 4; k++)
            {
                int lw = LevelDim(width, k);
                int lh = LevelDim(height, k);
                int stride = 1 << k;
                float[] low = new float[lw * lh];
                for (int j = 0; j < lh; j++)
                    for (int i = 0; i < lw; i++)
                        low[j * lw + i] = Linearize(stride * i, stride * j);   // top-left texel of each block
                lowDepth[k - 1] = low;
                int tw = LevelDim(width, k + 2);
                int th = LevelDim(height, k + 2);
                float pad = k <= 2 ? Linearize(width, height) : 0.0f;   // DS1:39-46 / DS2:35
                ushort[] atlas = new ushort[16 * tw * th];
                for (int s = 0; s < 16; s++)
                    for (int ty = 0; ty < th; ty++)
                        for (int tx = 0; tx < tw; tx++)
                        {
                            int i = 4 * tx + (s & 3);
                            int j = 4 * ty + (s >> 2);
                            float v = (i < lw && j < lh) ? low[j * lw + i] : pad;
                            atlas[(s * th + ty) * tw + tx] = ToHalfRtz(v);
                        }
                tiledDepth[k - 1] = atlas;
            }
        }

        // ---- pass 3: volumetric-obscurance render (REN main_interleaved) ------------------------
        static readonly int[] TermX = { 2, 4, 1, 2, 3, 1, 2 };
        static readonly int[] TermY = { 0, 0, 1, 2, 3, 3, 4 };
        static readonly int[] TermSlot = { 1, 3, 4, 8, 11, 6, 10 };    // REN:162-168

        ushort[] renAtlas;
        int renSw;
        int renSh;
        float renReject;

        float Tap(int s, int x, int y)
        {
            x = ClampI(x, 0, renSw - 1);
            y = ClampI(y, 0, renSh - 1);
            return FromHalf(renAtlas[(s * renSh + y) * renSw + x]);
        }

        float SamplePair(int s, int cx, int cy, int dx, int dy, float front, float invRange)   // REN:60-75
        {
            float d1 = Mad(Tap(s, cx + dx, cy + dy), invRange, -front);
            float d2 = Mad(Tap(s, cx - dx, cy - dy), invRange, -front);
            float p1 = Saturate(renReject * d1);
            float p2 = Saturate(renReject * d2);
            float sum = ClampF(d1, p2, 1.0f) + ClampF(d2, p1, 1.0f);
            return Saturate(Mad(-p1, p2, sum));
        }

        float Samples(int s, int cx, int cy, int x, int y, float invDepth, float invThickness)   // REN:77-110
        {
            float invRange = invThickness * invDepth;
            float front = invThickness - 0.5f;
            if (y == 0)
                return 0.5f * (SamplePair(s, cx, cy, x, 0, front, invRange) + SamplePair(s, cx, cy, 0, x, front, invRange));
            if (x == y)
                return 0.5f * (SamplePair(s, cx, cy, -x, x, front, invRange) + SamplePair(s, cx, cy, x, x, front, invRange));
            return 0.25f * (((SamplePair(s, cx, cy, x, y, front, invRange) + SamplePair(s, cx, cy, -x, y, front, invRange))
                             + SamplePair(s, cx, cy, y, x, front, invRange)) + SamplePair(s, cx, cy, -y, x, front, invRange));
        }

        void Render(int level)
        {
            // PushRenderCommands constant math (AO.cs:660-734)
            float[] thickness = new float[12];
            float[] fifth = { 0.0f, 0.2f, 0.4f, 0.6f, 0.8f };
            int[] tu = { 1, 2, 3, 4, 1, 1, 1, 1, 2, 2, 2, 3 };
            int[] tv = { 0, 0, 0, 0, 1, 2, 3, 4, 2, 3, 4, 3 };
            for (int i = 0; i < 12; i++)
            {
                float r = 1.0f - fifth[tu[i]] * fifth[tu[i]];
                if (tv[i] != 0) r = r - fifth[tv[i]] * fifth[tv[i]];
                thickness[i] = (float)Math.Sqrt((double)r);
            }
            renSw = LevelDim(width, level + 2);
            renSh = LevelDim(height, level + 2);
            float multiplier = 2.0f * (1.0f / projection00);
            multiplier = multiplier * 10.0f;
            multiplier = multiplier / (float)renSw;
            float inverseRange = 1.0f / multiplier;
            float[] invThickness = new float[12];
            float[] weight = new float[12];
            float[] count = { 0, 4, 0, 4, 4, 0, 8, 0, 4, 0, 8, 4 };
            float total = 0.0f;
            for (int i = 0; i < 12; i++)
            {
                invThickness[i] = inverseRange / thickness[i];
                weight[i] = count[i] == 0.0f ? 0.0f : count[i] * thickness[i];
                total += weight[i];
            }
            for (int i = 0; i < 12; i++) weight[i] /= total;
            renReject = -1.0f / thicknessModifier;
            renAtlas = tiledDepth[level - 1];

            int ow = LevelDim(width, level);
            int oh = LevelDim(height, level);
            byte[] dst = new byte[ow * oh];
            for (int Y = 0; Y < oh; Y++)
                for (int X = 0; X < ow; X++)
                {
                    int s = (X & 3) | ((Y & 3) << 2);       // REN:172 inverted
                    int cx = X >> 2;
                    int cy = Y >> 2;
                    float invDepth = 1.0f / Tap(s, cx, cy);
                    float ao = 0.0f;
                    for (int n = 0; n < 7; n++)
                        ao = Mad(weight[TermSlot[n]], Samples(s, cx, cy, TermX[n], TermY[n], invDepth, invThickness[TermSlot[n]]), ao);
                    dst[Y * ow + X] = ToUnorm8(Mad(intensity, ao - 1.0f, 1.0f));   // lerp(1, ao, gIntensity)
                }
            occlusion[level - 1] = dst;
        }

        // ---- pass 4: depth-aware blur + bilateral upsample (UPS) --------------------------------
        float upsStep;
        float upsBlurTol;

        bool CompareDeltas(float d1, float d2, float l1, float l2)      // UPS:83-87
        {
            float t = Mad(d1, d2, upsStep);
            return t * t > (l1 * l2) * upsBlurTol;
        }

        float SmartBlur5(float[] a, float[] z)                          // UPS:74-130, one 5-tap output
        {
            float d01 = z[1] - z[0];
            float d12 = z[2] - z[1];
            float d23 = z[3] - z[2];
            float d34 = z[4] - z[3];
            float l01 = Mad(d01, d01, upsStep);
            float l12 = Mad(d12, d12, upsStep);
            float l23 = Mad(d23, d23, upsStep);
            float l34 = Mad(d34, d34, upsStep);
            bool left = CompareDeltas(d01, d12, l01, l12);
            bool middle = CompareDeltas(d12, d23, l12, l23);
            bool right = CompareDeltas(d23, d34, l23, l34);
            float pc = a[2];
            float pb = (left || middle) ? a[1] : pc;
            float pa = left ? a[0] : pb;
            float pd = (right || middle) ? a[3] : pc;
            float pe = right ? a[4] : pd;
            return ((((pa + pe) * 0.5f + pb) + pc) + pd) * 0.25f;
        }

        byte[] Upsample(int lowLevel, float[] lowDepthTex, byte[] lowAo, float[] hiDepth32, ushort[] hiDepth16, byte[] hiAo)
        {
            int lw = LevelDim(width, lowLevel);
            int lh = LevelDim(height, lowLevel);
            int hw = LevelDim(width, lowLevel - 1);
            int hh = LevelDim(height, lowLevel - 1);
            // PushUpsampleCommands constant math (AO.cs:750-771)
            upsStep = 1920.0f / (float)lw;
            float bt = (float)Math.Pow(10.0, (double)blurTolerance) * upsStep;
            bt = 1.0f - bt;
            upsBlurTol = bt * bt;
            float tolerance = (float)Math.Pow(10.0, (double)upsampleTolerance);
            float noise = (float)Math.Pow(10.0, (double)noiseFilterTolerance) + tolerance;
            noise = 1.0f / noise;

            float[] inv = new float[lw * lh];
            float[] ao = new float[lw * lh];
            for (int i = 0; i < lw * lh; i++)
            {
                inv[i] = 1.0f / lowDepthTex[i];
                ao[i] = FromUnorm8(lowAo[i]);
            }
            int bw = lw + 2;                                  // blurred texels at virtual x = -1 .. lw
            float[] hb = new float[bw * lh];
            float[] a5 = new float[5];
            float[] z5 = new float[5];
            for (int y = 0; y < lh; y++)
                for (int vx = -1; vx <= lw; vx++)
                {
                    for (int t = 0; t < 5; t++)
                    {
                        int x = ClampI(vx - 2 + t, 0, lw - 1);
                        a5[t] = ao[y * lw + x];
                        z5[t] = inv[y * lw + x];
                    }
                    hb[y * bw + vx + 1] = SmartBlur5(a5, z5);
                }
            float[] vb = new float[bw * (lh + 2)];
            for (int r = 0; r < lh + 2; r++)
                for (int vx = -1; vx <= lw; vx++)
                {
                    int xc = ClampI(vx, 0, lw - 1);
                    for (int t = 0; t < 5; t++)
                    {
                        int y = ClampI(r - 1 - 2 + t, 0, lh - 1);
                        a5[t] = hb[y * bw + vx + 1];
                        z5[t] = inv[y * lw + xc];
                    }
                    vb[r * bw + vx + 1] = SmartBlur5(a5, z5);
                }

            int[] gx = { -1, 0, 0, -1 };                      // Gather order x, y, z, w as (col, row) offsets
            int[] gy = { 0, 0, -1, -1 };
            float[] num = { 9.0f, 3.0f, 1.0f, 3.0f };
            float[] w = new float[4];
            float[] la = new float[4];
            byte[] dst = new byte[hw * hh];
            for (int hy = 0; hy < hh; hy++)
                for (int hx = 0; hx < hw; hx++)
                {
                    int Dx = (hx + 1) >> 1;
                    int Dy = (hy + 1) >> 1;
                    int comp = (hx & 1) != 0 ? ((hy & 1) != 0 ? 3 : 0) : ((hy & 1) != 0 ? 2 : 1);   // UPS:229-232
                    float hiD = hiDepth32 != null ? hiDepth32[hy * hw + hx] : FromHalf(hiDepth16[hy * hw + hx]);
                    float hiA = hiAo != null ? FromUnorm8(hiAo[hy * hw + hx]) : 1.0f;
                    for (int t = 0; t < 4; t++)
                    {
                        int g = (comp + t) & 3;
                        int vx = Dx + gx[g];
                        int vy = Dy + gy[g];
                        float lo = lowDepthTex[ClampI(vy, 0, lh - 1) * lw + ClampI(vx, 0, lw - 1)];
                        w[t] = num[t] / (MathF.Abs(hiD - lo) + tolerance);       // UPS:179
                        la[t] = vb[(vy + 1) * bw + vx + 1];
                    }
                    float totalWeight = ((w[0] + w[1]) + w[2]) + w[3];
                    totalWeight = totalWeight + noise;
                    float sum = la[0] * w[0];
                    sum = Mad(la[1], w[1], sum);
                    sum = Mad(la[2], w[2], sum);
                    sum = Mad(la[3], w[3], sum);
                    sum = sum + noise;
                    dst[hy * hw + hx] = ToUnorm8((hiA * sum) / totalWeight);
                }
            return dst;
        }

        // ---- whole pipeline in the order of RebuildCommandBuffers (AO.cs:496-531) --------------
        public byte[] Run(float[] rawDepth, int pixelWidth, int pixelHeight)
        {
            width = pixelWidth;
            height = pixelHeight;
            depth = rawDepth;
            float fpn = farClipPlane / nearClipPlane;        // AO.cs:563
            if (usesReversedZBuffer) { zp0 = fpn - 1.0f; zp1 = 1.0f; }
            else { zp0 = 1.0f - fpn; zp1 = fpn; }
            Downsample();
            for (int level = 1; level <= 4; level++) Render(level);
            combined[2] = Upsample(4, lowDepth[3], occlusion[3], lowDepth[2], null, occlusion[2]);
            combined[1] = Upsample(3, lowDepth[2], combined[2], lowDepth[1], null, occlusion[1]);
            combined[0] = Upsample(2, lowDepth[1], combined[1], lowDepth[0], null, occlusion[0]);
            result = Upsample(1, lowDepth[0], combined[0], null, linearDepth, null);
            return result;
        }
    }
}
